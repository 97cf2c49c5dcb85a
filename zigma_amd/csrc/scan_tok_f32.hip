// Token-major selective-scan kernel, F32 I/O instantiation (see scan_tok.inc).
#include "scan_tok.inc"

namespace zigma {
int launch_scan_tok_f32(const zigma_scan_params_t &p, hipStream_t stream) { return launch_tok_io<F32>(p, stream); }
}  // namespace zigma
