// Library-level entry points of libfsgs.so (version, error reporting).
#include "common.h"

namespace fsgs {
thread_local int g_last_hip_error = 0;
}

extern "C" int fsgs_version(void) { return 100; /* 0.1.0 */ }

extern "C" int fsgs_abi_version(void) { return FSGS_ABI_VERSION; }

extern "C" int fsgs_grad_replica_lines(void) { return fsgs::kGradReplicas; }

extern "C" int fsgs_last_hip_error(void) { return fsgs::g_last_hip_error; }

extern "C" const char *fsgs_error_string(int code) {
    switch (code) {
        case FSGS_OK: return "ok";
        case FSGS_EINVAL: return "invalid argument (null pointer, unsupported size or channel count)";
        case FSGS_ELAUNCH: return "HIP launch/runtime error (see fsgs_last_hip_error)";
        case FSGS_ESCRATCH: return "scratch arena too small";
        case FSGS_EPROTOCOL: return "a bounded wait ran into its bound, or calls arrived out of order";
        default: return "unknown error";
    }
}
