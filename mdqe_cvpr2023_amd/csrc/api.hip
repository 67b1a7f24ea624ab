#include "common.h"
extern "C" int mdqe_version(void) { return 100; }
extern "C" int mdqe_abi_version(void) { return MDQE_ABI_VERSION; }
extern "C" const char* mdqe_strerror(int code) {
  switch (code) {
    case MDQE_OK: return "ok";
    case MDQE_EINVAL: return "invalid size or unsupported shape";
    case MDQE_ELAUNCH: return "kernel launch failed";
    case MDQE_ENULL: return "null pointer argument";
    case MDQE_ESTATE: return "object left inconsistent by an earlier failed call; create a new one";
    default: return "unknown error";
  }
}
