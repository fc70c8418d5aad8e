"""Development-only: does a stream created with hipExtStreamCreateWithCUMask restrict torch kernels to its CUs?
(Round 3: no -- 8192^3 f32 matmul 7.15-7.18 ms on every 1st / 2nd / 4th / 8th CU alike)"""
import ctypes, torch


def masked_stream(every, phase=0):
    hip = ctypes.CDLL("libamdhip64.so")
    words = (ctypes.c_uint32 * 10)()
    for cu in range(320):
        if every <= 1 or cu % every == phase % every:
            words[cu // 32] |= 1 << (cu % 32)
    handle = ctypes.c_void_p()
    err = hip.hipExtStreamCreateWithCUMask(ctypes.byref(handle), 10, words)
    if err != 0:
        raise RuntimeError("hipExtStreamCreateWithCUMask failed: %d" % err)
    return torch.cuda.ExternalStream(handle.value)


a = torch.randn(8192, 8192, device="cuda")
for every in (1, 2, 4, 8):
    st = masked_stream(every)
    with torch.cuda.stream(st):
        for _ in range(3):
            b = a @ a
        st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(5):
            b = a @ a
        e1.record(st)
        st.synchronize()
    print("every %d-th CU: 8192^3 f32 matmul %.2f ms" % (every, e0.elapsed_time(e1) / 5), flush=True)
