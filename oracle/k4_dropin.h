/* oracle/k4_dropin.h -- forced first include (g++ -include) for ngskit4b/KAligner.cpp and KAlignerCL.cpp when the
 * reference's own kalign front end is built on top of libk4sfx.so (make -C oracle ngskit4b_k4): INTEGRATION.md, option A.
 * libkit4b's class keeps existing under another name; from here on `CSfxArray` is the facade of include/k4_sfxarray.hpp
 * (renamed as well, so that its inline methods cannot collide at link time with libkit4b's CSfxArray::... symbols).
 * Test infrastructure: the result (oracle/_ref/ngskit4b_k4) exists to show that the reference's call sites compile and
 * run unchanged against the boundary, and to compare its SAM output with the CPU build's. */
#include <sys/mman.h>
#include <pthread.h>
#define CSfxArray CSfxArrayCPU
#include "libkit4b/commhdrs.h"
#undef CSfxArray
#define K4_HAVE_KIT4B_TYPES 1
#define CSfxArray CSfxArrayK4
#include "k4_sfxarray.hpp"
