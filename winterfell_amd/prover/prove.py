"""Prover::generate_proof (prover/src/lib.rs:275-492) for the built-in AIRs, every data-parallel step on the device.

Returns a `Proof` (prover/proof.py): the pieces air::proof::Proof is assembled from (context, commitments, out-of-domain
frame, FRI proof, proof-of-work nonce, the opened rows with their batch Merkle proofs) with `to_bytes()` in the reference's
wire format, plus the intermediate transcript values the tests compare."""
import ctypes
import time

from ..fri.prover import FriOptions, FriProver
from .channel import ProofOptions, ProverChannel
from .composer import DeepCompositionPoly, TracePolyTable, composition_poly_ood_frame
from .constraint_commitment import build_constraint_commitment
from .constraints import DefaultConstraintEvaluator
from .matrix import ColMatrix
from .proof import Proof
from .trace_lde import DefaultTraceLde, StarkDomain


def prove(air, trace: ColMatrix, options: ProofOptions, hasher, pub_inputs_elements, timings=None, build_aux_trace=None, transcript="host"):
    """build_aux_trace(aux_rand_elements) -> ColMatrix over E: Prover::build_aux_trace (prover/src/lib.rs:236-247), the user's
    builder of the auxiliary trace segment; needed exactly when the AIR is multi-segment.
    transcript = "device": the public coin lives on the device from the first commitment to the query positions
    (prove_device_transcript below: same proof, byte for byte, no host round trip in between); single-segment AIRs and the hashers
    with a device coin — anything else falls back to the host coin."""
    if transcript == "device" and not air.is_multi_segment() and hasher.DEVICE_COIN and options.grinding_factor <= MAX_DEVICE_GRIND and \
            FriOptions(options.blowup_factor, options.fri_folding_factor, options.fri_remainder_max_degree, field=air.FIELD).num_fri_layers(
                air.lde_domain_size()) > 0:
        return prove_device_transcript(air, trace, options, hasher, pub_inputs_elements, timings)
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


# the device search of the proof-of-work nonce is a fixed queue of launches over nonces 1 .. 2^(grinding_factor + 10): no nonce in that
# range has probability exp(-1024); above this factor the queue would be long, and prove() keeps the host loop (wf_grind)
MAX_DEVICE_GRIND = 24
DEVICE_GRIND_EXTRA_BITS = 10


def prove_device_transcript(air, trace: ColMatrix, options: ProofOptions, hasher, pub_inputs_elements, timings=None):
    """Prover::generate_proof (prover/src/lib.rs:282-492) with the Fiat-Shamir coin on the DEVICE for the whole transcript
    (prover/src/channel.rs:87-185): every reseed reads its digest from HBM, every draw leaves its elements in HBM, and the kernels that
    consume them (constraint evaluation, out-of-domain frames, DEEP composition, the FRI folds, the proof-of-work search, the query
    draw) read them there.  From the first commitment to the query positions nothing is copied to the host and the host waits for
    nothing: the whole chain is queued on the stream; then ONE wait, and the transcript comes back for the proof object.
    Same proof as prove(), byte for byte (tests/test_gpu_prove_device_transcript.py)."""
    import numpy as np
    import torch

    from .._lib import ptr
    from .composer import deep_compose_dev, ood_frame_dev
    from .constraints import ConstraintCompositionCoefficients, evaluate_constraints_dev
    f, ctx, D = air.FIELD, trace.ctx, options.ext_degree
    assert not air.is_multi_segment() and hasher.DEVICE_COIN
    assert trace.num_rows() == air.trace_length() and trace.num_cols() == air.TRACE_WIDTH
    tm = timings if timings is not None else {}
    ew = D * f.W
    t_all = time.perf_counter()
    channel = ProverChannel(air, options, hasher, pub_inputs_elements, ctx)      # seeds the coin: a host hash of the public inputs
    coin = channel.public_coin.to_device()
    domain = StarkDomain(air.trace_length(), options.blowup_factor, field=f)
    d_roots = ctx.empty_u8(2, 32)                                               # the two commitments, copied as the coin absorbs them
    # 1. main trace segment
    trace_lde, trace_polys = DefaultTraceLde.new(hasher, trace, domain, fetch_root=False)
    coin.reseed(trace_lde.main_segment_oracles.nodes_device[1], d_roots[0])
    # 2. constraint composition coefficients + evaluation
    nt, na = air.num_transition_constraints(), air.num_assertions()
    d_cc = coin.draw(D, nt + na)
    composition_poly_trace, assertions = evaluate_constraints_dev(air, trace_lde, domain, d_cc, D)
    # 3. constraint commitment
    constraint_commitment, composition_poly = build_constraint_commitment(
        hasher, composition_poly_trace, air.num_constraint_composition_columns(), domain, ext_degree=D, field=f, ctx=ctx, fetch_root=False)
    coin.reseed(constraint_commitment.vector_commitment.nodes_device[1], d_roots[1])
    # 4. out-of-domain point, frames, their digest into the coin, DEEP coefficients, DEEP composition
    d_z = coin.draw(D, 1)
    table = TracePolyTable(trace_polys)
    d_tframe = ood_frame_dev(trace_polys, d_z, D)                                # (2, trace columns, ew)
    d_qframe = ood_frame_dev(composition_poly.data, d_z, D)                      # (2, composition columns, ew)
    # merge_ood_evaluations (air/src/proof/ood_frame.rs:335-351): current rows (trace, quotient), then next rows
    merged = torch.cat([d_tframe[0].reshape(-1), d_qframe[0].reshape(-1), d_tframe[1].reshape(-1), d_qframe[1].reshape(-1)])
    d_ood_digest = ctx.empty_u8(1, 32)
    nel = merged.numel() // f.W
    ctx.call("wf_hash_elements_batch", hasher.HASH_ID, f.ID, ptr(merged), 1, nel, nel, ptr(d_ood_digest))
    coin.reseed(d_ood_digest)
    ntc, nqc = air.trace_width(), air.num_constraint_composition_columns()
    d_dcc = coin.draw(D, ntc + nqc)
    deep = deep_compose_dev(table, composition_poly, d_z, d_dcc, D)
    deep_evaluations = deep.evaluate(domain)
    # 5. FRI commit phase on the same coin (queued; its read-back is deferred)
    fri_options = FriOptions(options.blowup_factor, options.fri_folding_factor, options.fri_remainder_max_degree, field=f)
    fri_prover = FriProver(fri_options, hasher, ext_degree=D, ctx=ctx)
    ev = deep_evaluations.reshape(-1)
    length = ev.numel() // ew
    nl = fri_options.num_fri_layers(length)
    assert nl > 0, "a proof without FRI layers keeps the host transcript"
    off = f.element_words(int(fri_options.domain_offset()))
    finish_fri = fri_prover._build_layers_fused(channel, coin, ev, length, off.ctypes.data_as(ctypes.c_void_p), defer=True)
    # 6. proof of work and query positions, still on the device
    d_pow_seed = coin.state[:32].clone()
    d_nonce = ctx.empty_u64(1)
    ctx.call("wf_coin_grind", hasher.HASH_ID, ptr(coin.state), options.grinding_factor, options.grinding_factor + DEVICE_GRIND_EXTRA_BITS, ptr(d_nonce))
    d_pos = ctx.empty_u64(options.num_queries)
    ctx.call("wf_coin_draw_integers", hasher.HASH_ID, ptr(coin.state), ptr(d_nonce), options.num_queries, air.lde_domain_size().bit_length() - 1,
             ptr(d_pos))
    tm["queue_whole_transcript"] = (time.perf_counter() - t_all) * 1e3
    # ---- the one wait: the transcript comes back
    t0 = time.perf_counter()
    roots = ctx.to_host(d_roots)
    channel.commitments.append(np.array(roots[0], copy=True))
    channel.commitments.append(np.array(roots[1], copy=True))
    finish_fri()                                                                 # FRI roots, alphas, remainder (appends its commitments)
    coin.set_host_image(ctx.to_host(coin.state))
    coin.read()                                                                  # raises if a draw or the nonce search failed
    cc = ctx.to_host(d_cc).reshape(nt + na, ew)
    z = ctx.to_host(d_z).reshape(-1)
    tframe, qframe = ctx.to_host(d_tframe), ctx.to_host(d_qframe)
    dcc = ctx.to_host(d_dcc).reshape(ntc + nqc, ew)
    channel.pow_seed = ctx.to_host(d_pow_seed)
    channel.pow_nonce = int(ctx.to_host(d_nonce)[0])
    positions = [int(p) for p in ctx.to_host(d_pos)]
    query_positions = sorted(set(positions))
    channel.ood_frame = ((tframe[0], tframe[1]), (qframe[0], qframe[1]))
    deep.z, deep.cc_trace, deep.cc_constraints = z, dcc[:ntc], dcc[ntc:]
    tm["wait_and_read_transcript"] = (time.perf_counter() - t0) * 1e3
    # 7. openings (lib.rs:462-487)
    t0 = time.perf_counter()
    fri_layers, fri_remainder = fri_prover.layers, fri_prover.remainder_poly
    fri_proof = fri_prover.build_proof(query_positions)
    trace_queries = trace_lde.query(query_positions)
    constraint_queries = constraint_commitment.query(query_positions)
    ctx.sync()
    tm["build_proof_object"] = (time.perf_counter() - t0) * 1e3
    return Proof(air=air, hasher=hasher, options=options, commitments=channel.commitments, trace_commitment=roots[0],
                 constraint_commitment=roots[1], ood_point=z, ood_trace_frame=(tframe[0], tframe[1]),
                 ood_constraint_frame=(qframe[0], qframe[1]), constraint_coefficients=ConstraintCompositionCoefficients(cc[:nt], cc[nt:]),
                 assertions=assertions, deep_coefficients=(dcc[:ntc], dcc[ntc:]), fri_layers=fri_layers, fri_remainder=fri_remainder,
                 fri_proof=fri_proof, fri_alphas=channel.fri_alphas, fri_options=fri_options, pow_nonce=channel.pow_nonce, pow_seed=channel.pow_seed,
                 query_positions=query_positions, trace_queries=trace_queries, constraint_queries=constraint_queries,
                 num_composition_columns=composition_poly.num_columns(), aux_rand_elements=None, aux_trace_commitment=None, timings_ms=tm)
