import ctypes, numpy as np, time, os, sys
sys.path.insert(0, "/root/repo")
from jubjub_amd import _lib
_lib.load()
hip = ctypes.CDLL("libamdhip64.so.7") if False else ctypes.CDLL(None)
try:
    f = hip.hipHostRegister
except AttributeError:
    import importlib.util
    spec = importlib.util.find_spec("torch")
    hip = ctypes.CDLL(os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so"))
hip.hipHostRegister.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
hip.hipHostUnregister.argtypes = [ctypes.c_void_p]
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
for mb in (32, 96, 160):
    a = np.random.randint(0, 255, size=mb << 20, dtype=np.uint8)
    d = ctypes.c_void_p(); hip.hipMalloc(ctypes.byref(d), mb << 20)
    hip.hipMemcpy(d, a.ctypes.data, mb << 20, 1)
    t0 = time.perf_counter(); rc = hip.hipHostRegister(a.ctypes.data, mb << 20, 0); t1 = time.perf_counter()
    hip.hipMemcpy(d, a.ctypes.data, mb << 20, 1); t2 = time.perf_counter()
    rc2 = hip.hipHostUnregister(a.ctypes.data); t3 = time.perf_counter()
    hip.hipMemcpy(d, a.ctypes.data, mb << 20, 1); t4 = time.perf_counter()
    b = np.empty_like(a); t5 = time.perf_counter(); np.copyto(b, a); t6 = time.perf_counter()
    print("%d MB: register %.2f ms (rc %d), H2D pinned %.2f ms (%.1f GB/s), unregister %.2f ms, H2D pageable %.2f ms (%.1f GB/s), cpu memcpy %.2f ms" % (
        mb, (t1-t0)*1e3, rc, (t2-t1)*1e3, mb/1024/(t2-t1), (t3-t2)*1e3, (t4-t3)*1e3, mb/1024/(t4-t3), (t6-t5)*1e3))
