// f128 instantiation of the NTT engine (math/src/field/f128): 16-byte canonical elements, table twiddles.
#include "ntt_engine.cuh"

int wf_ntt_run_f128(wf_ctx *ctx, const NttJob &job) { return ntt_run<HostF128>(ctx, job); }
