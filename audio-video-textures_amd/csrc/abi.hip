// Error plumbing and device check of the C ABI (include/avt.h).
#include <stdarg.h>
#include <string.h>

#include "avt_common.h"

namespace avt {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace avt

extern "C" int avt_abi_version(void) { return AVT_ABI_VERSION; }
extern "C" const char* avt_last_error(void) { return avt::g_err; }

extern "C" int avt_device_check(char* name, size_t name_len) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) {
    avt::set_error("avt_device_check: hipGetDevice: %s", hipGetErrorString(e));
    return AVT_ERR_DEVICE;
  }
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) {
    avt::set_error("avt_device_check: hipGetDeviceProperties: %s", hipGetErrorString(e));
    return AVT_ERR_DEVICE;
  }
  if (name && name_len) {
    strncpy(name, prop.gcnArchName, name_len - 1);
    name[name_len - 1] = 0;
  }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    avt::set_error("avt_device_check: device %d is %s, this library is built for gfx950 only", dev, prop.gcnArchName);
    return AVT_ERR_DEVICE;
  }
  return AVT_OK;
}
