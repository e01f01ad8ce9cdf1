// libtedd_hip.so: the C ABI of include/tedd.h executed by hand-written HIP kernels for gfx950.
#include "tedd_hip_backend.hpp"
#define TEDD_BACKEND tedd::HipBackend
#include "tedd_api_impl.hpp"
