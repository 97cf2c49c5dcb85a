// Token-major selective-scan kernels, BF16 I/O instantiation (see scan_tok.inc, scan_tok2.inc).
#include "scan_tok.inc"
#include "scan_tok2.inc"

namespace zigma {
int launch_scan_tok_bf16(const zigma_scan_params_t &p, hipStream_t stream) {
    if ((tok2_eligible(p) || tok2_split_eligible(p)) && !(p.flags & ZIGMA_SCAN_PROBE_V1)) return launch_tok2<BF16>(p, stream);
    return launch_tok_io<BF16>(p, stream);
}
int launch_scan_tok_bf16_dtp(const zigma_scan_params_t &p, hipStream_t stream) {       // p.dt_x set: no other kernel serves it
    if (!(p.x ? tok2_dtp_split_ok(p) : tok2_dtp_ok(p)) || (p.flags & ZIGMA_SCAN_PROBE_V1)) return ZIGMA_ERR_UNSUPPORTED;
    return launch_tok2<BF16>(p, stream);
}
}  // namespace zigma
