"""How long pinning caller memory takes (hipHostRegister) and how fast copies from pinned vs pageable memory run."""
import ctypes, time, numpy as np
hip = ctypes.CDLL("libamdhip64.so")
hip.hipHostRegister.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
hip.hipHostUnregister.argtypes = [ctypes.c_void_p]
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
n = 256 << 20
a = np.ones(n, dtype=np.uint8)
d = ctypes.c_void_p()
assert hip.hipMalloc(ctypes.byref(d), n) == 0
for rep in range(2):
    t = time.perf_counter(); assert hip.hipMemcpy(d, a.ctypes.data, n, 1) == 0; t1 = time.perf_counter() - t
    t = time.perf_counter(); assert hip.hipMemcpy(a.ctypes.data, d, n, 2) == 0; t2 = time.perf_counter() - t
    print("pageable H2D %.1f GB/s  D2H %.1f GB/s" % (n / t1 * 1e-9, n / t2 * 1e-9))
t = time.perf_counter(); rc = hip.hipHostRegister(a.ctypes.data, n, 0); tr = time.perf_counter() - t
print("register rc", rc, "%.1f ms (%.1f GB/s equivalent)" % (tr * 1e3, n / tr * 1e-9))
for rep in range(2):
    t = time.perf_counter(); assert hip.hipMemcpy(d, a.ctypes.data, n, 1) == 0; t1 = time.perf_counter() - t
    t = time.perf_counter(); assert hip.hipMemcpy(a.ctypes.data, d, n, 2) == 0; t2 = time.perf_counter() - t
    print("registered H2D %.1f GB/s  D2H %.1f GB/s" % (n / t1 * 1e-9, n / t2 * 1e-9))
t = time.perf_counter(); hip.hipHostUnregister(a.ctypes.data); print("unregister %.1f ms" % ((time.perf_counter() - t) * 1e3))
