// tests/emul/hip/hip_ext.h -- TEST-ONLY stand-in (see hip_runtime.h in this directory)
#pragma once
#include "hip_runtime.h"
