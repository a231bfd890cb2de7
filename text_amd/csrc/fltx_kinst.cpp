/*
 * fltx_kinst.cpp -- one group of decode-kernel instantiations (see
 * fltx_instances.h), selected with -DFLTX_INST_W=<threads> -DFLTX_INST_G=<group>.
 * __graft_entry__.build compiles one object per (size, group) in parallel and
 * links them all into text_amd/lib/libfltx.so.
 */
#if !defined(FLTX_INST_W) || !defined(FLTX_INST_G)
#error "compile with -DFLTX_INST_W=<64|128|256|512|1024> -DFLTX_INST_G=<1..10>"
#endif
#include "fltx_kernel_entry.h"

#define FLTX_INST(...) template __global__ void __VA_ARGS__(DecodeParams);
#include "fltx_instances.h"
#undef FLTX_INST
