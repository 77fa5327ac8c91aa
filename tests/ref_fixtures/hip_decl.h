// hip_decl.h -- TEST INFRASTRUCTURE: the `_hip` functions, declared for the reference's fixtures in the reference's own types.
//
// Every line HOOK(pointer, variant) of svt-av1-psy_amd/csrc/rtcd_hooks.def (the list csrc/rtcd_hook.hip installs from) becomes
//     extern "C" <function type of the reference's dispatch pointer> variant;
// so a fixture that takes `svt_nxm_sad_kernel_helper_avx2` takes `svt_nxm_sad_kernel_hip` with no cast: the function type is read off the reference's
// declaration of the pointer (Source/Lib/Codec/aom_dsp_rtcd.h, common_dsp_rtcd.h), not retyped here.
#ifndef SVT_HIP_FIXTURE_DECL_H
#define SVT_HIP_FIXTURE_DECL_H
#include <type_traits>

#include "definitions.h"
#include "aom_dsp_rtcd.h"
#include "common_dsp_rtcd.h"

#define HOOK(ptr, fn) extern "C" std::remove_pointer_t<decltype(ptr)> fn;
#include "../../svt-av1-psy_amd/csrc/rtcd_hooks.def"
#undef HOOK

#endif
