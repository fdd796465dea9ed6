/* shim so that #include "osqp_api_types.h" (reference bindings.cpp.in:10) resolves to this engine's C ABI */
#include "../osqp_hip.h"
