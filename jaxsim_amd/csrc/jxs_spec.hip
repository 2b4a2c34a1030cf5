// One model-specialised kernel (include/jaxsim_amd.h, "model-specialised kernels"): the kernel source of
// jxs_kernels.h / jxs_core.h compiled for ONE (dtype, lanes per environment, mode) with the integer model flags
// of JXS_SPEC_ASSIGN as constants.  Built by jaxsim_amd/specialize.py:
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -fvisibility=hidden -Djxs_launch=jxs_launch_spec
//         -DJXS_SPEC_T=float -DJXS_SPEC_G=32 -DJXS_SPEC_MODE=0 '-DJXS_SPEC_ASSIGN=P.nL=24,...' '-DJXS_SPEC_STRING="..."'
// The namespace rename and the hidden visibility keep the host stubs of this object apart from the generic
// ones of libjaxsim_amd.so (the same template instantiation would otherwise be bound to the first definition).
#include "jxs_kernels.h"

#if !defined(JXS_SPEC_T) || !defined(JXS_SPEC_G) || !defined(JXS_SPEC_MODE) || !defined(JXS_SPEC_ASSIGN) || !defined(JXS_SPEC_STRING)
#error "jxs_spec.hip is compiled by jaxsim_amd/specialize.py with the JXS_SPEC_* definitions"
#endif

extern "C" __attribute__((visibility("default"))) const char* jxs_spec_string() { return JXS_SPEC_STRING; }

extern "C" __attribute__((visibility("default"))) int jxs_spec_launch(const void* params, const unsigned char* mblk, const void* args, void* stream) {
  return (int)jxs_launch::launch_one<JXS_SPEC_T, JXS_SPEC_G, JXS_SPEC_MODE>(*static_cast<const jxs::KParams<JXS_SPEC_T>*>(params), mblk,
                                                                         *static_cast<const jxs::KArgs<JXS_SPEC_T>*>(args),
                                                                         static_cast<hipStream_t>(stream));
}
