import ctypes as C, os, torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libtr_probe.so"))
lib.run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
def probe(addr_fn, title):
    addr = torch.tensor([addr_fn(l) for l in range(64)], dtype=torch.int32, device="cuda")
    out = torch.zeros(256, dtype=torch.int16, device="cuda")
    lib.run(addr.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
    o = out.view(64, 4).cpu().tolist()
    print(title)
    for l in range(0, 64, 1):
        if l < 20 or l % 16 == 0: print("  lane", l, "addr(elems)", addr_fn(l) // 2, "->", o[l])
# pattern 1: lane l reads row l (row pitch 64 elements = 128 B), 4 consecutive elements at column 0
probe(lambda l: l * 128, "P1: lane l -> 4 elems at row l (pitch 64 elems), col 0")
# pattern 2: 16-lane group g, lane i: row = i % 4 ... try [4 rows][16 cols] block: lane i -> row i//4?? use addr = (i%16)*8 bytes contiguous
probe(lambda l: (l % 16) * 8 + (l // 16) * 512, "P2: lane i -> 4 contiguous elems at offset 4*(i%16) (+256 elems per 16-lane group)")
