/* compat/fieldio.hpp -- lets a translation unit that says `#include "fieldio.hpp"` (main.cpp:18) build against the MI355X
 * engine with -I<repo>/xlab-fftbarotropic_amd/host/compat -I<repo>/include and -lfieldio (or -lfftbaro). */
#include "fieldio_fb.h"
