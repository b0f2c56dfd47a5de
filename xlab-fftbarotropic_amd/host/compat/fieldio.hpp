// compat/fieldio.hpp -- the reference's declarations (fieldio.hpp:5-6); the definitions are in lib/libfieldio.so
// and libfftbaro.so (csrc/fb_fieldio.cpp), exported with the reference's mangled names.
#include <cstddef>
#ifndef FIELDIO_H
#define FIELDIO_H
void writeField(const char *filename, float *data, size_t len);
void readField(const char *filename, float *data, size_t len);
#endif
