/* Plain-C consumer of include/sdt_hip.h: proves the boundary is a C ABI (no C++ / torch types in the signatures).
 * Runs without a GPU: only the argument-validation paths are exercised (they return before any launch). */
#include <stdio.h>
#include <string.h>

#include "sdt_hip.h"

int main(void) {
    sdt_conv_geom g;
    int rc;
    memset(&g, 0, sizeof g);
    if (sdt_abi_version() != 5) return 10;
    if (sizeof(sdt_conv_geom) != (17 + 3 * SDT_MAX_TAPS) * sizeof(int32_t)) return 11;
    if (sizeof(sdt_wt_desc) != 4 * sizeof(void*) + 6 * sizeof(int32_t)) return 12;
    rc = sdt_conv_taps_f32(NULL, NULL, NULL, NULL, &g, NULL); /* zero geometry -> argument error, message set */
    if (rc != SDT_ERR_ARG || strlen(sdt_last_error()) == 0) return 13;
    printf("last error: %s\n", sdt_last_error());
    rc = sdt_adam_step_f32(NULL, NULL, NULL, NULL, 0, NULL, 0.9f, 0.999f, 1e-8f, 0.f, 1.f, NULL, NULL);
    if (rc != SDT_ERR_ARG) return 14;
    if (sdt_set_conv_math(42) != SDT_ERR_ARG || sdt_get_conv_math() != SDT_MATH_F32) return 15;
    if (sdt_set_conv_math(SDT_MATH_BF16) != SDT_OK || sdt_get_conv_math() != SDT_MATH_BF16) return 16;
    if (sdt_set_conv_math(SDT_MATH_F32) != SDT_OK) return 17;
    /* the bf16-storage entry points: a NULL geometry is unsupported (0 / -1), a NULL tensor an argument error */
    if (sdt_convsk_supported_t(NULL, 1, SDT_BF16) != 0 || sdt_convsk_plan_bytes_t(NULL, 1, SDT_BF16) != -1) return 18;
    if (sdt_convsk_dw_supported_t(NULL, SDT_BF16) != 0) return 19;
    rc = sdt_convsk_bf16(NULL, NULL, NULL, NULL, NULL, NULL, NULL, 1u, NULL, NULL, 0, 0, 0, NULL);
    if (rc != SDT_ERR_ARG) return 20;
    rc = sdt_colnorm_fwd_t(NULL, SDT_BF16, NULL, SDT_BF16, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, 1, 1, 4, 1e-5f, 0.1f, 0.2f, 0, NULL);
    if (rc != SDT_ERR_ARG) return 21;
    if (sdt_convsk_set_spin_limit(sdt_convsk_get_spin_limit()) != SDT_OK) return 22;
    /* the Conv1d chain: the wiring is validated on the host before anything is launched */
    {
        sdt_chain1d_layer L[2];
        float dummy = 0.f;
        unsigned words[2] = {0u, 0u};
        memset(L, 0, sizeof L);
        if (sizeof(sdt_chain1d_layer) != 10 * sizeof(int32_t) + 6 * sizeof(void*)) return 23;
        L[0].Ti = 64, L[0].To = 64, L[0].Cin = 288, L[0].k = 3, L[0].stride = 1, L[0].pad = 1, L[0].in_mode = SDT_CHAIN_PLAIN, L[0].src_a = L[0].src_b = -1;
        L[1] = L[0];
        L[1].Cin = 256, L[1].in_mode = SDT_CHAIN_NORM, L[1].src_a = 1; /* a block cannot read itself */
        L[0].w = L[1].w = &dummy, L[0].y = L[1].y = &dummy;
        rc = sdt_chain1d_fwd_f32(L, 2, &dummy, &dummy, 4, 0.2f, 1e-5f, SDT_MATH_F32, words, words + 1, NULL);
        if (rc != SDT_ERR_ARG) return 24;
        L[1].src_a = 0, L[1].k = 5, L[1].pad = 2; /* no K loop was built for a 5-tap block */
        if (sdt_chain1d_supported(L, 2, 4) != 0) return 25;
        if (sdt_chain1d_bwd_f32(NULL, 2, &dummy, 4, 0.2f, 1e-5f, 1, SDT_MATH_F32, words, words + 1, NULL) != SDT_ERR_ARG) return 26;
    }
    puts("C ABI OK");
    return 0;
}
