"""Prover::generate_proof (prover/src/lib.rs:275-492) for the built-in AIRs, every data-parallel step on the device.

Returns a `Proof` (prover/proof.py): the pieces air::proof::Proof is assembled from (context, commitments, out-of-domain
frame, FRI proof, proof-of-work nonce, the opened rows with their batch Merkle proofs) with `to_bytes()` in the reference's
wire format, plus the intermediate transcript values the tests compare."""
import time

from ..fri.prover import FriOptions, FriProver
from .channel import ProofOptions, ProverChannel
from .composer import DeepCompositionPoly, TracePolyTable, composition_poly_ood_frame
from .constraint_commitment import build_constraint_commitment
from .constraints import DefaultConstraintEvaluator
from .matrix import ColMatrix
from .proof import Proof
from .trace_lde import DefaultTraceLde, StarkDomain


def prove(air, trace: ColMatrix, options: ProofOptions, hasher, pub_inputs_elements, timings=None, build_aux_trace=None):
    """build_aux_trace(aux_rand_elements) -> ColMatrix over E: Prover::build_aux_trace (prover/src/lib.rs:236-247), the user's
    builder of the auxiliary trace segment; needed exactly when the AIR is multi-segment."""
    f, ctx, D = air.FIELD, trace.ctx, options.ext_degree
    assert (build_aux_trace is not None) == air.is_multi_segment(), "a multi-segment AIR comes with Prover::build_aux_trace, the others without"
    assert trace.num_rows() == air.trace_length() and trace.num_cols() == air.TRACE_WIDTH
    tm = timings if timings is not None else {}

    def lap(name, t0):
        ctx.sync()
        tm[name] = tm.get(name, 0.0) + (time.perf_counter() - t0) * 1e3

    channel = ProverChannel(air, options, hasher, pub_inputs_elements, ctx)
    domain = StarkDomain(air.trace_length(), options.blowup_factor, field=f)
    # 1. commit to the main trace segment (lib.rs:305-306, 500-533)
    t0 = time.perf_counter()
    trace_lde, trace_polys = DefaultTraceLde.new(hasher, trace, domain)
    channel.commit_trace(trace_lde.get_main_trace_commitment())
    lap("commit_to_main_trace_segment", t0)
    # 1b. the auxiliary trace segment (lib.rs:320-346): draw its random elements, build it, extend + commit
    aux_rand_elements = aux_polys = aux_commitment = None
    if air.is_multi_segment():
        t0 = time.perf_counter()
        aux_rand_elements = channel.get_aux_rand_elements()
        aux_trace = build_aux_trace(aux_rand_elements)
        assert aux_trace.num_cols() == air.AUX_TRACE_WIDTH and aux_trace.ext_degree == D and aux_trace.num_rows() == air.trace_length()
        aux_polys, aux_commitment = trace_lde.set_aux_trace(aux_trace, domain)
        channel.commit_trace(aux_commitment)
        lap("commit_to_aux_trace_segment", t0)
    # 2. evaluate constraints (lib.rs:353-364)
    t0 = time.perf_counter()
    evaluator = DefaultConstraintEvaluator(air, channel.get_constraint_composition_coeffs(), D, aux_rand_elements=aux_rand_elements)
    composition_poly_trace = evaluator.evaluate(trace_lde, domain)
    lap("evaluate_constraints", t0)
    # 3. commit to the constraint evaluations (lib.rs:366-371, 535-575)
    t0 = time.perf_counter()
    constraint_commitment, composition_poly = build_constraint_commitment(
        hasher, composition_poly_trace, air.num_constraint_composition_columns(), domain, ext_degree=D, field=f, ctx=ctx)
    channel.commit_constraints(constraint_commitment.commitment())
    lap("commit_to_constraint_evaluations", t0)
    # 4. out-of-domain frames and the DEEP composition polynomial (lib.rs:373-431)
    t0 = time.perf_counter()
    z = channel.get_ood_point()
    table = TracePolyTable(trace_polys)
    if aux_polys is not None:
        table.add_aux_segment(aux_polys)                                              # lib.rs:341
    ood_trace_states = table.get_ood_frame(z, D)
    ood_evaluations = composition_poly_ood_frame(composition_poly, z, D)
    channel.send_ood_evaluations(ood_trace_states, ood_evaluations)
    cc_trace, cc_constraints = channel.get_deep_composition_coeffs()
    deep = DeepCompositionPoly(z, cc_trace, cc_constraints, D)
    deep.add_trace_polys(table, composition_poly, ood_trace_states, ood_evaluations)
    assert deep.degree() == air.trace_length() - 2                                   # lib.rs:423
    deep_evaluations = deep.evaluate(domain)
    lap("build_and_evaluate_deep_composition_poly", t0)
    # 5. FRI commit phase (lib.rs:433-440)
    t0 = time.perf_counter()
    fri_options = FriOptions(options.blowup_factor, options.fri_folding_factor, options.fri_remainder_max_degree, field=f)
    fri_prover = FriProver(fri_options, hasher, ext_degree=D, ctx=ctx)
    fri_prover.build_layers(channel, deep_evaluations)
    lap("compute_fri_layers", t0)
    # 6. proof of work + query positions (lib.rs:444-459)
    t0 = time.perf_counter()
    channel.grind_query_seed()
    query_positions = channel.get_query_positions()
    lap("determine_query_positions", t0)
    # 7. openings (lib.rs:462-487)
    t0 = time.perf_counter()
    fri_layers, fri_remainder = fri_prover.layers, fri_prover.remainder_poly      # kept for callers that inspect the layers
    fri_proof = fri_prover.build_proof(query_positions)
    trace_queries = trace_lde.query(query_positions)
    constraint_queries = constraint_commitment.query(query_positions)
    lap("build_proof_object", t0)
    return Proof(air=air, hasher=hasher, options=options, commitments=channel.commitments, trace_commitment=trace_lde.get_main_trace_commitment(),
                 constraint_commitment=constraint_commitment.commitment(), ood_point=z, ood_trace_frame=ood_trace_states,
                 ood_constraint_frame=ood_evaluations, constraint_coefficients=evaluator.cc, assertions=evaluator.assertions,
                 deep_coefficients=(cc_trace, cc_constraints), fri_layers=fri_layers, fri_remainder=fri_remainder, fri_proof=fri_proof,
                 fri_alphas=channel.fri_alphas, fri_options=fri_options, pow_nonce=channel.pow_nonce, pow_seed=channel.pow_seed, query_positions=query_positions,
                 trace_queries=trace_queries, constraint_queries=constraint_queries, num_composition_columns=composition_poly.num_columns(),
                 aux_rand_elements=aux_rand_elements, aux_trace_commitment=aux_commitment,
                 timings_ms=tm)
