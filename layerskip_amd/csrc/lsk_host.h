// Host-side plumbing shared by every translation unit of the product library (lsk_engine.hip, lsk_generate.hip) and of the test
// library (lsk_test_exports.hip): the error convention of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <mutex>

#include "lsk_common.h"

// each library defines these once (the message buffer is thread-local)
int lsk_fail(const char* fmt, ...);

#define HIP_OK(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) return lsk_fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

#define LSK_TRY(expr)                \
    do {                             \
        int _r = (expr);             \
        if (_r != 0) return _r;      \
    } while (0)
