"""ProverChannel (prover/src/channel.rs:17-215): the prover's side of the Fiat-Shamir transcript.

Host logic around a DefaultRandomCoin; the one data-parallel step, grind_query_seed, runs on the GPU.  The coin is seeded
exactly like the reference's: hash_elements(Context::to_elements() ++ PublicInputs::to_elements()) (channel.rs:57-75) with
the reference's encodings of TraceInfo, the field modulus, the constraint count and ProofOptions, so every value drawn from
it — and therefore every commitment, out-of-domain frame, FRI layer, nonce and query position of prove() — is the one the
Rust prover produces for the same trace and options (tests/test_gpu_proof_artefacts.py compares them with the CPU oracle's
restatement of the whole prover)."""
import numpy as np

from ..crypto.random import DefaultRandomCoin, grind_query_seed
from .constraints import ConstraintCompositionCoefficients


class ProofOptions:
    """air::ProofOptions (air/src/options.rs:88-118), the fields the prover pipeline needs."""

    def __init__(self, num_queries, blowup_factor, grinding_factor, ext_degree=1, fri_folding_factor=4, fri_remainder_max_degree=31):
        assert 0 < num_queries <= 255 and blowup_factor >= 2 and blowup_factor & (blowup_factor - 1) == 0 and grinding_factor <= 32
        self.num_queries, self.blowup_factor, self.grinding_factor = num_queries, blowup_factor, grinding_factor
        self.ext_degree, self.fri_folding_factor, self.fri_remainder_max_degree = ext_degree, fri_folding_factor, fri_remainder_max_degree


def trace_info_to_elements(main_width, trace_length, element_bytes, aux_width=0, num_aux_rands=0, meta=b""):
    """TraceInfo::to_elements (air/src/air/trace_info.rs:209-238), canonical integers: segment widths packed 8 bits each
    into one element, the trace length, then the metadata in chunks of ELEMENT_BYTES - 1 bytes."""
    num_aux_segments = 1 if aux_width else 0
    buf = (main_width << 8) | num_aux_segments
    if num_aux_segments:
        buf = (((buf << 8) | aux_width) << 8) | num_aux_rands
    out = [buf, trace_length & 0xFFFFFFFF]
    step = element_bytes - 1
    return out + [int.from_bytes(meta[i:i + step], "little") for i in range(0, len(meta), step)]


def proof_options_to_elements(options):
    """ProofOptions::to_elements (air/src/options.rs:294-305): FieldExtension discriminants are None = 1, Quadratic = 2,
    Cubic = 3, i.e. the extension degree."""
    buf = (((((options.ext_degree << 8) | options.fri_folding_factor) << 8) | options.fri_remainder_max_degree) << 8) | options.blowup_factor
    return [buf, options.grinding_factor, options.num_queries]


def context_to_elements(air, options, aux_width=None, num_aux_rands=None, meta=b""):
    """Context::to_elements (air/src/proof/context.rs:106-137) for a built-in AIR, as canonical integers: trace info, the
    field modulus' little-endian bytes split into two elements, the number of constraints, the proof options."""
    f = air.FIELD
    nbytes = 8 * f.W
    aux_width = air.AUX_TRACE_WIDTH if aux_width is None else aux_width
    num_aux_rands = air.NUM_AUX_RANDS if num_aux_rands is None else num_aux_rands
    out = trace_info_to_elements(air.TRACE_WIDTH, air.trace_length(), nbytes, aux_width, num_aux_rands, meta)
    mb = f.M.to_bytes(nbytes, "little")
    out += [int.from_bytes(mb[:nbytes // 2], "little"), int.from_bytes(mb[nbytes // 2:], "little")]
    out.append((air.num_assertions() + air.num_transition_constraints()) & 0xFFFFFFFF)
    return out + proof_options_to_elements(options)


class ProverChannel:
    def __init__(self, air, options: ProofOptions, hasher, pub_inputs_elements, ctx=None):
        """pub_inputs_elements: PublicInputs::to_elements() of the AIR in the field's internal representation (rescue:
        seed ++ result, examples/src/rescue/air.rs:45-51; the Fibonacci examples: the result; vdf: [seed, result])."""
        f = air.FIELD
        self.air, self.options, self.hasher, self.ctx = air, options, hasher, ctx
        self.context_elements = context_to_elements(air, options)
        seed = f.pack([f.new(v) for v in self.context_elements] + list(pub_inputs_elements))
        self.public_coin = DefaultRandomCoin(hasher, f, seed, ctx)
        self.commitments, self.fri_alphas = [], []
        self.ood_frame = None
        self.pow_nonce, self.pow_seed = 0, None

    # ---- commitments (channel.rs:87-110)
    def commit_trace(self, trace_root):
        self.commitments.append(np.array(trace_root, copy=True))
        self.public_coin.reseed(trace_root)

    def commit_constraints(self, constraint_root):
        self.commitments.append(np.array(constraint_root, copy=True))
        self.public_coin.reseed(constraint_root)

    def send_ood_evaluations(self, trace_ood_frame, constraints_ood_frame):
        """merge_ood_evaluations (air/src/proof/ood_frame.rs:335-351): current rows (trace, quotient), then next rows."""
        (tc, tn), (qc, qn) = trace_ood_frame, constraints_ood_frame
        self.ood_frame = (trace_ood_frame, constraints_ood_frame)
        evals = np.concatenate([np.asarray(tc).reshape(-1), np.asarray(qc).reshape(-1), np.asarray(tn).reshape(-1), np.asarray(qn).reshape(-1)])
        self.public_coin.reseed(self.hasher.hash_elements(evals, self.ctx, field=self.air.FIELD))

    # ---- public coin (channel.rs:112-165); linear batching: one draw per coefficient (air/src/air/coefficients.rs:201-206)
    def get_aux_rand_elements(self):
        """Air::get_aux_rand_elements(channel.public_coin()) (air/src/air/mod.rs:292-306; prover/src/lib.rs:323-325): the random
        elements the auxiliary trace segment is built from, drawn once the main segment is committed"""
        return self.public_coin.draw_many(self.air.NUM_AUX_RANDS, self.options.ext_degree)

    def get_constraint_composition_coeffs(self):
        D = self.options.ext_degree
        self.public_coin.prefetch(self.air.num_transition_constraints() + self.air.num_assertions())
        t = self.public_coin.draw_many(self.air.num_transition_constraints(), D)
        b = self.public_coin.draw_many(self.air.num_assertions(), D)
        return ConstraintCompositionCoefficients(t, b)

    def get_ood_point(self):
        return self.public_coin.draw(self.options.ext_degree)

    def get_deep_composition_coeffs(self):
        D = self.options.ext_degree
        self.public_coin.prefetch(self.air.trace_width() + self.air.num_constraint_composition_columns())
        trace = self.public_coin.draw_many(self.air.trace_width(), D)            # TraceInfo::width: main + auxiliary columns
        constraints = self.public_coin.draw_many(self.air.num_constraint_composition_columns(), D)
        return trace, constraints

    # ---- fri::ProverChannel (fri/src/prover/channel.rs:24-50)
    def commit_fri_layer(self, layer_root):
        self.commitments.append(np.array(layer_root, copy=True))
        self.public_coin.reseed(layer_root)

    def draw_fri_alpha(self):
        a = self.public_coin.draw(self.options.ext_degree)
        self.fri_alphas.append(a)
        return a

    # the same two calls for ALL the FRI layers with the coin on the device (FriProver.build_layers' fused loop)
    def fri_device_coin(self):
        """the public coin, handed to the device for the layer loop; None when the hasher does not suit (hash.py DEVICE_COIN)"""
        return self.public_coin.to_device() if self.hasher.DEVICE_COIN else None

    def absorb_fri_layers(self, device_coin, roots, alphas, remainder_commitment=None):
        """what commit_fri_layer / draw_fri_alpha would have recorded layer by layer, then the coin back on the host"""
        for root, alpha in zip(roots, alphas):
            self.commitments.append(np.array(root, copy=True))
            self.fri_alphas.append(np.array(alpha, copy=True))
        if remainder_commitment is not None:                    # the remainder's commit_fri_layer happened on the device as well
            self.commitments.append(np.array(remainder_commitment, copy=True))
        self.public_coin.take_back(device_coin)

    # ---- query phase (channel.rs:146-185)
    def grind_query_seed(self):
        self.pow_seed = np.array(self.public_coin.seed, copy=True)
        self.pow_nonce = grind_query_seed(self.hasher, self.public_coin.seed, self.options.grinding_factor, ctx=self.ctx)

    def get_query_positions(self):
        positions = self.public_coin.draw_integers(self.options.num_queries, self.air.lde_domain_size(), self.pow_nonce)
        return sorted(set(positions))
