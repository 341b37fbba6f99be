// odin_internal.h -- host-side helpers shared by the translation units of libodin_hip.so
#pragma once
#include <cstdio>
#include <cstring>
#include "../../include/odin_hip.h"

#define ODIN_MAX_SLAB_BLOCKS 256

int odin_fail(int code, const char* msg);
int odin_check_launch(const char* what);
int odin_num_cus();
