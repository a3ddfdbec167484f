// f62 instantiation of the NTT engine (math/src/field/f62): Montgomery u64 words, table twiddles.
#include "ntt_engine.cuh"

int wf_ntt_run_f62(wf_ctx *ctx, const NttJob &job) { return ntt_run<HostF62>(ctx, job); }
