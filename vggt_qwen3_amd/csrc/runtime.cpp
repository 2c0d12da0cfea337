// Host-side runtime bits of libvq3hip.so: ABI version, per-thread error string.
#include <cstdarg>
#include <cstdio>

#include "vq3_hip.h"

namespace {
thread_local char g_err[512] = {0};
}

void vq3_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* vq3_last_error(void) { return g_err; }
extern "C" int vq3_abi_version(void) { return VQ3_ABI_VERSION; }
extern "C" const char* vq3_target_arch(void) { return "gfx950"; }
