// error.cpp — thread-local last-error text behind dabhip_last_error().
#include <string>

#include "../../include/dabhip.h"

namespace dabhip {
static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
}  // namespace dabhip

extern "C" const char* dabhip_last_error(void) { return dabhip::g_last_error.c_str(); }
