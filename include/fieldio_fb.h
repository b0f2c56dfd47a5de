/*
 * fieldio_fb.h -- the two field I/O functions of XLab-FFTBarotropic under their C++ names, served by the engine
 * (csrc/fb_fieldio.cpp -> lib/libfieldio.so and libfftbaro.so).  Replaces the reference's fieldio.hpp:5-6 / lib/libfieldio.so:
 * the exported symbols are the same mangled names (_Z10writeFieldPKcPfm, _Z9readFieldPKcPfm), the files are raw host-order
 * float32 with no header (fieldio.cpp:7-33), the stderr lines are the reference's.  C callers use fb_write_field / fb_read_field
 * (include/fftbaro.h), which also return a status.
 */
#ifndef FIELDIO_FB_H
#define FIELDIO_FB_H
#include <cstddef>

void readField(const char *path, float *host_buffer, std::size_t n_floats);     /* fieldio.cpp:21-33 */
void writeField(const char *path, float *host_buffer, std::size_t n_floats);    /* fieldio.cpp:7-19  */

#endif /* FIELDIO_FB_H */
