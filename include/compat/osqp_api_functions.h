/* shim so that #include "osqp_api_functions.h" (reference bindings.cpp.in:9) resolves to this engine's C ABI */
#include "../osqp_hip.h"
