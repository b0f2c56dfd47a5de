/* compat/fftw3.h -- lets a translation unit that says `#include <fftw3.h>` (main.cpp:12) build against the
 * MI355X engine with -I<repo>/xlab-fftbarotropic_amd/host/compat -I<repo>/include and -lfftw3f_fb. */
#include "fftw3_fb.h"
