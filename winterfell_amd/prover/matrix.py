"""ColMatrix / RowMatrix (prover/src/matrix/col_matrix.rs, row_matrix.rs) with HBM-resident data."""
import ctypes

import numpy as np

from .._lib import WF_FIELD_F64, default_context, load_library, ptr
from ..crypto.merkle import MerkleTree
from ..math import fields


class PartitionOptions:
    """air::PartitionOptions (air/src/options.rs:405-451)."""

    def __init__(self, num_partitions=1, hash_rate=1):
        assert 1 <= num_partitions <= 16, "number of partitions must be in [1, 16]"
        assert 1 <= hash_rate <= 256, "hash rate must be in [1, 256]"
        self.num_partitions = num_partitions
        # the reference stores `hash_rate as u8` (air/src/options.rs:414-418), so the permitted value 256 wraps to 0
        self.hash_rate = hash_rate & 0xFF

    def partition_size(self, num_columns, ext_degree=1):
        if self.num_partitions == 1:
            return num_columns
        return max(-(-num_columns // self.num_partitions), self.hash_rate // ext_degree)

    def num_partitions_for(self, num_columns, ext_degree=1):
        return -(-num_columns // self.partition_size(num_columns, ext_degree))


class ColMatrix:
    """Column-major matrix: `data` is a (num_cols, num_rows * ext_degree) device tensor (one contiguous column per row
    of the tensor).  ColMatrix::new asserts: at least one column, power-of-two length (col_matrix.rs:44-62)."""

    def __init__(self, columns, ext_degree=1, ctx=None, field=fields.f64):
        self.ctx = ctx or default_context()
        self.ext_degree = ext_degree
        self.field = field
        data = self.ctx.to_device(columns) if isinstance(columns, np.ndarray) else columns
        assert data.dim() == 2 and data.shape[0] > 0, "a matrix must contain at least one column"
        n = data.shape[1] // (ext_degree * field.W)
        assert n > 1 and n & (n - 1) == 0, "number of rows in a matrix must be a power of 2 greater than 1"
        self.data = data

    def num_cols(self):
        return self.data.shape[0]

    def num_base_cols(self):
        return self.data.shape[0] * self.ext_degree

    def num_rows(self):
        return self.data.shape[1] // (self.ext_degree * self.field.W)

    def col_stride(self):
        return self.data.shape[1] // self.field.W      # in base elements

    def interpolate_columns(self):
        """ColMatrix::interpolate_columns (col_matrix.rs:192-202): returns a new matrix of coefficients."""
        out = self.data.clone()
        log_n = self.num_rows().bit_length() - 1
        self.ctx.call("wf_interpolate_columns", self.field.ID, self.ext_degree, ptr(out), self.num_cols(), self.col_stride(), log_n)
        return ColMatrix(out, self.ext_degree, self.ctx, self.field)

    def evaluate_columns_over(self, domain):
        """ColMatrix::evaluate_columns_over (col_matrix.rs:230-243): the polynomials in the columns evaluated over the LDE
        coset, column-major (the layout benches/row_matrix.rs compares RowMatrix against)."""
        f, D = self.field, self.ext_degree
        n, b = self.num_rows(), domain.trace_to_lde_blowup()
        assert n == domain.trace_length
        out = self.ctx.empty_u64(self.num_cols(), n * b * D * f.W)
        off = f.element_words(int(domain.offset))
        self.ctx.call("wf_evaluate_columns_over", f.ID, D, ptr(self.data), self.num_cols(), self.col_stride(), n.bit_length() - 1,
                      b.bit_length() - 1, off.ctypes.data_as(ctypes.c_void_p), ptr(out), n * b * D)
        return ColMatrix(out, D, self.ctx, f)

    def hash_rows(self, hasher):
        leaves = self.ctx.empty_u8(self.num_rows(), 32)
        self.ctx.call("wf_hash_columns", hasher.HASH_ID, self.field.ID, self.ext_degree, ptr(self.data), self.num_cols(), self.col_stride(),
                      self.num_rows(), ptr(leaves))
        return leaves

    def commit_to_rows(self, hasher):
        """ColMatrix::commit_to_rows (col_matrix.rs:262-286): leaf r = hash_elements(row r), then the vector commitment."""
        return MerkleTree.new(hasher, self.hash_rows(hasher), self.ctx)

    # ---- element access (col_matrix.rs:85-165); host round trips, for tests and small fix-ups
    def get(self, col_idx, row_idx):
        assert col_idx < self.num_cols() and row_idx < self.num_rows()
        w = self.ext_degree * self.field.W
        return self.ctx.to_host(self.data[col_idx, row_idx * w:(row_idx + 1) * w])

    def get_column(self, col_idx):
        return self.ctx.to_host(self.data[col_idx])

    def read_row_into(self, row_idx):
        assert row_idx < self.num_rows()
        w = self.ext_degree * self.field.W
        return self.ctx.to_host(self.data[:, row_idx * w:(row_idx + 1) * w]).reshape(-1)

    def merge_column(self, column):
        """col_matrix.rs:150-155: append a column (its length must match)."""
        col = self.ctx.to_device(np.ascontiguousarray(column, dtype=np.uint64).reshape(1, -1)) if isinstance(column, np.ndarray) else column.reshape(1, -1)
        assert col.shape[1] == self.data.shape[1], "column length must match the matrix"
        self.data = _cat([self.data, col])

    def remove_column(self, index):
        assert index < self.num_cols(), "column index out of range"
        col = self.data[index].clone()
        self.data = _cat([self.data[:index], self.data[index + 1:]])
        return col

    def to_host(self):
        return self.ctx.to_host(self.data)


def _cat(parts):
    import torch
    return torch.cat(parts, dim=0).contiguous()


class RowMatrix:
    """Row-major matrix (prover/src/matrix/row_matrix.rs:28-41): data[row * row_width + col], row_width =
    8 * ceil(base_cols / 8), only the first elements_per_row words of a row are meaningful."""

    def __init__(self, data, row_width, elements_per_row, ext_degree, ctx, field=fields.f64):
        self.data, self.row_width, self.elements_per_row, self.ext_degree, self.ctx = data, row_width, elements_per_row, ext_degree, ctx
        self.field = field

    @classmethod
    def evaluate_polys_over(cls, polys: ColMatrix, blowup, domain_offset):
        """RowMatrix::evaluate_polys_over::<8> (row_matrix.rs:84-100)."""
        ctx = polys.ctx
        n = polys.num_rows()
        log_n, log_b = n.bit_length() - 1, blowup.bit_length() - 1
        assert blowup & (blowup - 1) == 0
        f = polys.field
        rw = load_library().wf_row_width(polys.num_cols(), polys.ext_degree)
        out = ctx.empty_u64(n * blowup, rw * f.W)
        off = f.element_words(int(domain_offset))
        ctx.call("wf_evaluate_polys_over", f.ID, polys.ext_degree, ptr(polys.data), polys.num_cols(), polys.col_stride(),
                 log_n, log_b, off.ctypes.data_as(ctypes.c_void_p), ptr(out))
        return cls(out, rw, polys.num_base_cols(), polys.ext_degree, ctx, f)

    @classmethod
    def evaluate_polys(cls, polys: ColMatrix, blowup_factor):
        """RowMatrix::evaluate_polys (row_matrix.rs:57-74): the same over the coset with offset B::GENERATOR."""
        return cls.evaluate_polys_over(polys, blowup_factor, polys.field.new(polys.field.GENERATOR))

    def num_rows(self):
        return self.data.shape[0]

    def num_cols(self):
        return self.elements_per_row // self.ext_degree

    def get(self, col_idx, row_idx):
        """row_matrix.rs:154-158"""
        w = self.ext_degree * self.field.W
        assert col_idx < self.num_cols() and row_idx < self.num_rows()
        return self.ctx.to_host(self.data[row_idx, col_idx * w:(col_idx + 1) * w])

    def row(self, idx):
        assert idx < self.num_rows()
        return self.ctx.to_host(self.data[idx, : self.elements_per_row * self.field.W])

    def rows(self, positions):
        """Batch row fetch for TraceLde::query (wf_rows_fetch)."""
        pos = np.ascontiguousarray(positions, dtype=np.uint64)
        out = np.empty((len(pos), self.elements_per_row * self.field.W), dtype=np.uint64)
        self.ctx.call("wf_rows_fetch", ptr(self.data), self.row_width, self.elements_per_row, 8 * self.field.W,
                      pos.ctypes.data_as(ctypes.c_void_p), len(pos), out.ctypes.data_as(ctypes.c_void_p))
        return out

    def hash_rows(self, hasher, partition_options=None):
        po = partition_options or PartitionOptions()
        leaves = self.ctx.empty_u8(self.num_rows(), 32)
        self.ctx.call("wf_hash_rows", hasher.HASH_ID, self.field.ID, self.ext_degree, ptr(self.data), self.num_rows(), self.row_width,
                      self.elements_per_row, po.num_partitions, po.hash_rate or 256, ptr(leaves))
        return leaves

    def commit_to_rows(self, hasher, partition_options=None):
        """RowMatrix::commit_to_rows (row_matrix.rs:184-228) -> MerkleTree."""
        return MerkleTree.new(hasher, self.hash_rows(hasher, partition_options), self.ctx)

    def to_host(self):
        return self.ctx.to_host(self.data)
