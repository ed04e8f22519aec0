#include <cstdlib>
// Library-wide state: last-error buffer, cached device properties, version.
#include "rpb_common.h"

thread_local char rpb_err_buf[512] = "";

extern "C" const char* rpb_last_error() { return rpb_err_buf; }
extern "C" int rpb_bf16_const_planes() { return RPB_BF16_CONST_PLANES; }   // planes of the fp32 constants multiplied with bf16-STORED operands
extern "C" int rpb_abi_version() { return 2; }          // == RPB_ABI_VERSION of include/rpb.h (tests/test_abi.py compares them)

void rpb_cmx_claim_prealloc();
static int g_line_claim_mode = -1;
int rpb_line_claim_mode() {
    if (g_line_claim_mode < 0) {
        const char* e = getenv("RPB_LINE_CLAIM");
        g_line_claim_mode = e ? atoi(e) : 1;
        if (g_line_claim_mode < 0 || g_line_claim_mode > 2) g_line_claim_mode = 1;
    }
    return g_line_claim_mode;
}
extern "C" int rpb_line_claim_set(int mode) {          // 0 / 1 / 2, or -1: back to RPB_LINE_CLAIM / the default
    if (mode < -1 || mode > 2) RPB_FAIL(RPB_ERR_ARG, "line_claim_set: mode %d (0 static deal, 1 workgroup counter, 2 chip-wide counter, -1 default)", mode);
    g_line_claim_mode = mode;
    if (mode == 2) rpb_cmx_claim_prealloc();            // the counter ring of the current device (rpb_cmx.hip): allocated here, not in a launch
    return RPB_OK;
}

int rpb_num_cus() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cus[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        // RPB_RESERVE_CUS = r (data-parallel runs): every persistent kernel of the library launches n - r workgroups (they take one CU each:
        // two waves per SIMD at 256 registers), which leaves r CUs to the collective kernels of the side stream -- otherwise a collective's
        // kernel waits for a compute workgroup to retire (up to a kernel duration, ~1-3 ms) or, once running, delays the next compute
        // kernel's last workgroups (both regimes were measured with the modelled transfers).  Read ONCE per process:
        // partial-row counts handed to Python (rpb_*_rows / _slots) and launch grids must agree for the life of the workspaces.
        const char* e = getenv("RPB_RESERVE_CUS");
        int r = e ? atoi(e) : 0;
        if (r < 0) r = 0;
        if (r > n / 2) r = n / 2;
        cus[dev] = n - r;
    }
    return cus[dev];
}
