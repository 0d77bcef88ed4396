/* Test hooks of libsed_hip.so -- NOT part of the product ABI (include/sed_hip.h).  They exist so that tests/test_gpu_gru.py can
 * drive the fused GRU recurrence (sed_gru_seq_fwd / sed_gru_seq_bwd, replacing nn.GRU of reference pytorch/models.py:529-530,
 * :565-567) into its rarely taken paths on demand.  Bound by `_lib.test_hooks()` only; nothing under the package's product
 * modules may call them (tests/test_capi_and_host.py checks). */
#ifndef SED_HIP_TEST_H
#define SED_HIP_TEST_H
#include "sed_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* polls before a workgroup of the fused recurrence gives up waiting for its partners (default 2^23, about 1 s; <= 0 restores
 * the default) */
int sed_gru_set_spin_limit(long spins);
/* the workgroups of a group normally find themselves on one XCD (they check the XCC_ID register) and exchange through its L2
 * with plain stores; 1 makes them use the agent-scope stores of the fallback path */
int sed_gru_force_agent_scope(int on);
/* holds `blocks` CUs (one workgroup with lds_bytes of LDS each) for `microseconds` */
int sed_debug_occupy(int blocks, int lds_bytes, long microseconds, sed_stream_t stream);
/* the slice reduce of sed_conv3x3_wgrad_sf16 alone (tools/wgrad_reduce_bench.py: which cut for which slice count), partial
 * [nparts][9][Cout][Cin]; variant 0 = the library's choice, 1 = 64 elements x 4 part groups, 2 = x 16 part groups, 3 = rows form */
int sed_test_wgrad_sf16_reduce(const float* partial, int nparts, int Cout, int Cin, const float* gy_amax, const float* x_amax,
                               float* dw_oihw, int variant, sed_stream_t stream);
#ifdef __cplusplus
}
#endif
#endif
