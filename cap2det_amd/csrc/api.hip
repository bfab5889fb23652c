// Version / error strings of the C-ABI (include/cap2det_hip.h).
#include "c2d_common.h"
#include <stdlib.h>
#include <string>
#include <utility>
#include <vector>

namespace {
const std::vector<std::pair<std::string, std::string>>& tune_table() {
  static const std::vector<std::pair<std::string, std::string>> t = [] {
    std::vector<std::pair<std::string, std::string>> out;
    const char* e = getenv("C2D_TUNE");
    std::string s = e ? e : "";
    size_t pos = 0;
    while (pos < s.size()) {
      size_t end = s.find(',', pos);
      if (end == std::string::npos) end = s.size();
      const std::string item = s.substr(pos, end - pos);
      const size_t eq = item.find('=');
      if (eq != std::string::npos) out.emplace_back(item.substr(0, eq), item.substr(eq + 1));
      pos = end + 1;
    }
    return out;
  }();
  return t;
}
}  // namespace

static int g_available_cus = 256;
int c2d_available_cus() { return g_available_cus; }
extern "C" int c2d_set_available_cus(int cus) {
  C2D_CHECK_ARG(cus >= 8 && cus <= 256);
  g_available_cus = cus;
  return C2D_OK;
}
extern "C" int c2d_get_available_cus(void) { return g_available_cus; }

bool c2d_tune_on() { return getenv("C2D_TUNE") != nullptr; }
const char* c2d_tune_get(const char* key) {
  for (const auto& kv : tune_table())
    if (kv.first == key) return kv.second.c_str();
  return nullptr;
}

extern "C" int c2d_version(void) { return C2D_ABI_VERSION; }

extern "C" const char* c2d_error_string(int code) {
  switch (code) {
    case C2D_OK: return "ok";
    case C2D_ERR_INVALID_ARG: return "invalid argument";
    case C2D_ERR_UNSUPPORTED: return "unsupported configuration";
    case C2D_ERR_LAUNCH: return "kernel launch failed";
    case C2D_ERR_WORKSPACE: return "workspace too small";
    case C2D_ERR_DATA: return "malformed input data";
    default: return "unknown error";
  }
}
