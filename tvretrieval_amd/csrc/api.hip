// ABI bookkeeping for libxmlhip.so.
#include "common.h"

extern "C" int xml_abi_version(void) {
  return 6; }
extern "C" const char* xml_build_arch(void) { return "gfx950"; }
extern "C" const char* xml_status_string(int status) {
  switch (status) {
    case XML_OK: return "ok";
    case XML_ERR_BAD_ARG: return "bad argument";
    case XML_ERR_UNSUPPORTED: return "unsupported shape";
    case XML_ERR_WORKSPACE: return "workspace too small";
    case XML_ERR_LAUNCH: return "kernel launch failed";
    default: return "unknown status";
  }
}
