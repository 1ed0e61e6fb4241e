// Version / status strings of the C ABI (include/trx.h).
#include "trx_common.h"

extern "C" int trx_version(void) { return TRX_VERSION; }

extern "C" const char *trx_status_string(int status)
{
    switch (status) {
    case TRX_OK: return "ok";
    case TRX_ERR_ARG: return "invalid argument (null pointer, non-positive size or bad enum)";
    case TRX_ERR_NDIM: return "unsupported dimensionality (ndim must be 2 or 3; D must be 1 when ndim is 2)";
    case TRX_ERR_WORKSPACE: return "workspace too small (query trx_*_workspace_bytes)";
    case TRX_ERR_HIP: return "HIP kernel launch failed";
    case TRX_ERR_CAPACITY: return "loss-curve buffer shorter than the requested iterations";
    default: return "unknown trx status";
    }
}
