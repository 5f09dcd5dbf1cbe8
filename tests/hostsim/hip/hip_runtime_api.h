// FAKE <hip/hip_runtime_api.h> - TEST INFRASTRUCTURE ONLY: fq_comm.cpp on the SIMT-emulator build
#pragma once
#include "hip_runtime.h"
