// FAKE <rccl/rccl.h> - TEST INFRASTRUCTURE ONLY (tests/hostsim): the types and enums fq_comm.cpp needs to compile
// against the fake HIP runtime.  The functions themselves come from whatever library FASTP_GPU_RCCL_LIB names
// (tests/rccl_stub); nothing here is linked.
#pragma once
#include <stddef.h>
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclInt = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;
