// One translation unit per (dtype, mode): explicit instantiation of the launcher (and through it of the
// kernels for every lane-group size).  build.sh compiles this file 16 times in parallel.
#include "jxs_kernels.h"

#ifndef JXS_INST_T
#error "compile with -DJXS_INST_T=float|double -DJXS_INST_MODE=<jxs::Mode>"
#endif

namespace jxs_launch {
template hipError_t launch_g<JXS_INST_T, JXS_INST_MODE>(int, const jxs::KParams<JXS_INST_T>&, const unsigned char*, const jxs::KArgs<JXS_INST_T>&, hipStream_t);
}
