// fl_abi.h -- what every extern "C" entry point of the library has in common.
#pragma once
#include <new>

#include "../../include/fanlin_gpu.h"

// No exception may cross the C ABI (the reference's caller turns any Err of process_image into its fallback image / 500,
// src/main.rs:185-195; an exception through extern "C" would abort the server): the entry points that allocate are
// function-try-blocks ending in this.
#define FL_ABI_CATCH catch (const std::bad_alloc &) { return FLGPU_ERR_OOM; } catch (...) { return FLGPU_ERR_DEVICE; }
