/* Plain-C consumer of include/sdt_hip.h: proves the boundary is a C ABI (no C++ / torch types in the signatures).
 * Runs without a GPU: only the argument-validation paths are exercised (they return before any launch). */
#include <stdio.h>
#include <string.h>

#include "sdt_hip.h"

int main(void) {
    sdt_conv_geom g;
    int rc;
    memset(&g, 0, sizeof g);
    if (sdt_abi_version() != 3) return 10;
    if (sizeof(sdt_conv_geom) != (17 + 3 * SDT_MAX_TAPS) * sizeof(int32_t)) return 11;
    if (sizeof(sdt_wt_desc) != 4 * sizeof(void*) + 4 * sizeof(int32_t)) return 12;
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
    puts("C ABI OK");
    return 0;
}
