import sys, ctypes as C
sys.path.insert(0, sys.argv[1])
import torch, xlab_fftbarotropic_amd as X
L = X.lib()
for n in (4096, 8192, 16384, 2048):
    m = X.Model(n, n)
    xl, ks, ky0, e = C.c_int(), C.c_int(), C.c_int(), C.c_size_t()
    L.fb_slab_geometry(m.fop._h, C.byref(xl), C.byref(ks), C.byref(ky0), C.byref(e))
    print("n=%d pitch=%d (16*%d, base %d)" % (n, ks.value, ks.value // 16, (n // 2 + 1 + 15) // 16 * 16))
    m.close()
