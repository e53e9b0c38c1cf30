import ctypes, torch
hip = ctypes.CDLL("libamdhip64.so")
torch.zeros(1, device="cuda")
def q(stream, n):
    m = (ctypes.c_uint32 * n)(*([0xdeadbeef] * n))
    rc = hip.hipExtStreamGetCUMask(ctypes.c_void_p(stream), n, m)
    return rc, [hex(x) for x in m]
s_cur = torch.cuda.current_stream().cuda_stream
s_new = torch.cuda.Stream().cuda_stream
ms = ctypes.c_void_p(); mask = (ctypes.c_uint32 * 8)(0xffffffff, 0, 0, 0, 0, 0, 0, 0)
print("create", hip.hipExtStreamCreateWithCUMask(ctypes.byref(ms), 8, mask))
for name, s in (("current", s_cur), ("new", s_new), ("masked", ms.value)):
    for n in (8, 16, 4):
        print(name, n, q(s, n))
