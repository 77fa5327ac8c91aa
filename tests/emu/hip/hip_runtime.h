// test-only shim: lets `#include <hip/hip_runtime.h>` resolve to the CPU lock-step interpreter
// when the kernel sources are built with g++ -Itests/emu (see tests/emu/hipemu.h).
#pragma once
#include "../hipemu.h"
