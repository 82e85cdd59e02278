// (temporary) launcher stubs until mlp16_bwd.hip / mlp16_dw.hip land
#include "mcnerf_16.h"
hipError_t mcn16_launch_bwd(const Mcn16BwdArgs&, hipStream_t) { return hipErrorNotSupported; }
hipError_t mcn16_launch_dw(const Mcn16DwArgs&, hipStream_t) { return hipErrorNotSupported; }
