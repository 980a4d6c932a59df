import ctypes as C, time
L = C.CDLL("tests/hip/libexact_check.so")
for w, name in ((0, "sqrt"), (1, "rcp"), (2, "rcp_sqrt"), (3, "raw rcp (control)"), (4, "raw sqrt (control)")):
    r = (C.c_ulonglong * 2)()
    t = time.time()
    e = L.exact_check_run(w, r)
    print(name, "err", e, "mismatches", r[0], "first", hex(r[1]), f"{time.time()-t:.2f}s", flush=True)
