"""air::proof::Proof and its byte serialisation (air/src/proof/mod.rs:56-225): what `prove()` returns, assembled from the
device-resident pieces — the proof context, the commitments, the queried trace / constraint rows with their batch Merkle
openings, the out-of-domain frame, the FRI proof and the proof-of-work nonce — in the reference's own wire format, so that
`Proof.to_bytes()` is what `winterfell::Proof::to_bytes()` returns for the same trace and options.

Encodings follow the reference's `Serializable` impls one for one:
  ByteWriter::write_usize       utils/core/src/serde/byte_writer.rs:77-91,145-149  (variable-length: 1..9 bytes)
  Vec<T>                        usize length prefix, then the elements               utils/core/src/serde/mod.rs:295-299
  field elements                canonical little-endian integers (f64 8 bytes, f128 16); extension = its base elements
  digests                       Digest::as_bytes (32 raw bytes; 24 for Blake3_192; 4 canonical words for the Rescue digests)
  TraceInfo                     air/src/air/trace_info.rs:240-263      ProofOptions   air/src/options.rs:307-321
  Context                       air/src/proof/context.rs:139-152       Commitments    air/src/proof/commitments.rs:100-112
  Queries                       air/src/proof/queries.rs:51-73,138-146 OodFrame       air/src/proof/ood_frame.rs:59-108,178-188
  BatchMerkleProof              crypto/src/merkle/proofs.rs:390-400    FriProof / FriProofLayer  fri/src/proof.rs:198-213,343-353
"""
import numpy as np

BATCHING_LINEAR = 0            # air::BatchingMethod::Linear (air/src/options.rs:479-483): what prove() implements


def write_usize(value):
    """ByteWriter::write_usize: the vint64 encoding (a length byte count in the low bits of the first byte)."""
    value = int(value)
    zeros = 64 - value.bit_length()
    length = 9 - min(max(zeros - 1, 0) // 7, 8)
    if length == 9:
        return b"\x00" + value.to_bytes(8, "little")
    return ((((value << 1) | 1) << (length - 1)) & ((1 << 64) - 1)).to_bytes(8, "little")[:length]


def elements_to_bytes(field, words):
    """write_many over field elements: canonical little-endian bytes of every base element, in order."""
    w = np.ascontiguousarray(words, dtype=np.uint64).reshape(-1)
    if field.W == 1:
        from ..math import fields as _f
        if field is _f.f64:
            return _f.to_ints(w).tobytes()
        return np.array([field.as_int(int(v) % field.M) for v in w], dtype=np.uint64).tobytes()
    return w.tobytes()                      # f128 keeps canonical integers: the words are the little-endian bytes


def batch_proof_to_bytes(hasher, proof):
    """BatchMerkleProof::write_into: depth (u8), the number of node vectors (usize), then every vector as a Vec<Digest>."""
    out = bytearray([proof.depth])
    out += write_usize(len(proof.nodes))
    for nodes in proof.nodes:
        out += write_usize(len(nodes))
        for d in nodes:
            out += hasher.digest_as_bytes(d)
    return bytes(out)


def queries_to_bytes(field, hasher, rows, proof):
    """Queries::new + write_into: the queried rows as one Vec<u8> of element bytes, then the opening proof as a Vec<u8>."""
    values = elements_to_bytes(field, rows)
    paths = batch_proof_to_bytes(hasher, proof)
    return write_usize(len(values)) + values + write_usize(len(paths)) + paths


def trace_info_to_bytes(main_width, trace_length, aux_width=0, num_aux_rands=0, meta=b""):
    return bytes([main_width, aux_width, num_aux_rands, trace_length.bit_length() - 1]) + len(meta).to_bytes(2, "little") + meta


def proof_options_to_bytes(options, num_partitions=1, hash_rate=1):
    return bytes([options.num_queries, options.blowup_factor, options.grinding_factor, options.ext_degree, options.fri_folding_factor,
                  options.fri_remainder_max_degree, BATCHING_LINEAR, BATCHING_LINEAR, num_partitions, hash_rate & 0xFF])


def context_to_bytes(air, options):
    f = air.FIELD
    modulus = f.M.to_bytes(8 * f.W, "little")
    return (trace_info_to_bytes(air.TRACE_WIDTH, air.trace_length(), air.AUX_TRACE_WIDTH, air.NUM_AUX_RANDS) + bytes([len(modulus)]) + modulus +
            proof_options_to_bytes(options) +
            write_usize(air.num_assertions() + air.num_transition_constraints()))


def ood_frame_to_bytes(field, trace_frame, quotient_frame):
    """OodFrame: for each of the two parts a u16 byte count, then [frame size = 2 (u8), current row, next row]."""
    out = bytearray()
    for cur, nxt in (trace_frame, quotient_frame):
        states = bytes([2]) + elements_to_bytes(field, cur) + elements_to_bytes(field, nxt)
        out += len(states).to_bytes(2, "little") + states
    return bytes(out)


def fri_proof_to_bytes(field, hasher, fri_proof):
    out = bytearray([len(fri_proof.layers)])
    for layer in fri_proof.layers:
        values = elements_to_bytes(field, layer.values)
        paths = batch_proof_to_bytes(hasher, layer.proof)
        out += len(values).to_bytes(4, "little") + values + len(paths).to_bytes(4, "little") + paths
    rem = elements_to_bytes(field, fri_proof.remainder)
    out += len(rem).to_bytes(2, "little") + rem
    out += bytes([fri_proof.num_partitions().bit_length() - 1])
    return bytes(out)


class Proof:
    """What prove() produced, by the names of air::proof::Proof's fields where they exist, plus the intermediate values the
    tests compare (coefficients, alphas, ...)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def to_bytes(self):
        """Proof::to_bytes (air/src/proof/mod.rs:189-199)."""
        air, f, h = self.air, self.air.FIELD, self.hasher
        out = bytearray(context_to_bytes(air, self.options))
        out.append(len(self.query_positions))                                               # num_unique_queries
        com = b"".join(h.digest_as_bytes(c) for c in self.commitments)                      # trace roots, constraint root, FRI roots
        out += len(com).to_bytes(2, "little") + com
        for rows, (_, bp) in self.trace_queries:                                            # one Queries per trace segment
            out += queries_to_bytes(f, h, rows, bp)
        c_rows, (_, c_bp) = self.constraint_queries
        out += queries_to_bytes(f, h, c_rows, c_bp)
        out += ood_frame_to_bytes(f, self.ood_trace_frame, self.ood_constraint_frame)
        out += fri_proof_to_bytes(f, h, self.fri_proof)
        out += int(self.pow_nonce).to_bytes(8, "little")
        return bytes(out)
