#include <cstdlib>
// Library-wide state: last-error buffer, cached device properties, version.
#include "rpb_common.h"

thread_local char rpb_err_buf[512] = "";

extern "C" const char* rpb_last_error() { return rpb_err_buf; }
extern "C" int rpb_abi_version() { return 1; }

int rpb_line_claim_mode() {
    static int mode = -1;
    if (mode < 0) {
        const char* e = getenv("RPB_LINE_CLAIM");
        mode = e ? atoi(e) : 1;
        if (mode < 0 || mode > 2) mode = 1;
    }
    return mode;
}

int rpb_num_cus() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cus[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev] = n;
    }
    return cus[dev];
}
