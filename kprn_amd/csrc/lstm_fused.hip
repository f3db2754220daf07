// Fused persistent LSTM kernels (placeholder until the MFMA kernels land): reports
// "unsupported" so the engine takes the generic GEMM pipeline.
#include "kprn_internal.h"
namespace fused {
bool fwd_supported(const kprn_handle*, int) { return false; }
void forward(kprn_handle*, const kprn_batch*, bool) {}
bool bwd_supported(const kprn_handle*, int) { return false; }
void backward(kprn_handle*, const kprn_batch*, int) {}
void params_changed(kprn_handle*) {}
void release(kprn_handle*) {}
}  // namespace fused
