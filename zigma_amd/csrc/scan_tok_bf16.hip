// Token-major selective-scan kernels, BF16 I/O instantiation (see scan_tok.inc, scan_tok2.inc).
#include <stdlib.h>
#include <string.h>

#include "scan_tok.inc"
#include "scan_tok2.inc"

namespace zigma {
int launch_scan_tok_bf16(const zigma_scan_params_t &p, hipStream_t stream) {
    const char *k = getenv("ZIGMA_SCAN_KERNEL");        // A/B knob for tools/scan_ab.py: "v1" pins the first-generation kernel
    if ((tok2_eligible(p) || tok2_split_eligible(p)) && !(k && strcmp(k, "v1") == 0)) return launch_tok2<BF16>(p, stream);
    return launch_tok_io<BF16>(p, stream);
}
}  // namespace zigma
