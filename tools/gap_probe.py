"""What the queue leaves between two dependent launches, as a function of what the first one wrote (round 6):
    cd /tmp && rocprofv3 --kernel-trace -d <dir> -o g -- python tools/gap_probe.py run      then      python tools/gap_probe.py read <db>
`run`: for several sizes and store kinds, kernel A (k_gap_store: streams X MB) followed by kernel B (k_gap_touch: one word), 20 times.
`read`: from the trace, the median distance from A's end to B's start per (size, kind), and A's median duration."""
import ctypes as C, os, sqlite3, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
SIZES_MB = [1, 8, 32, 128]
MODES = {0: 'plain', 1: 'nontemporal', 2: 'sc0 sc1', 3: 'loads only'}
REP = 20

if sys.argv[1] == 'run':
    import torch
    from bundlesdf_amd import build
    so = C.CDLL(build.build_probe(verbose=False))
    so.nof_gap_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    word = torch.zeros(16, device='cuda')
    src = torch.randn(128 * 2 ** 20 // 4, device='cuda')
    dst = torch.empty_like(src)
    st = torch.cuda.current_stream().cuda_stream
    for mb in SIZES_MB:
        for mode in MODES:
            for _ in range(REP):
                rc = so.nof_gap_probe(dst.data_ptr(), src.data_ptr(), mb * 2 ** 20 // 16, mode, 2048, word.data_ptr(), st)
                assert rc == 0
            torch.cuda.synchronize()
    # the same pairs (128 MB, plain stores: gapless above) with a timing event recorded on the stream after every pair
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(REP)]
    for i in range(REP):
        rc = so.nof_gap_probe(dst.data_ptr(), src.data_ptr(), 128 * 2 ** 20 // 16, 0, 2048, word.data_ptr(), st)
        evs[i].record()
    torch.cuda.synchronize()
    print('done', word[0].item())
else:
    import statistics
    rows = sqlite3.connect(sys.argv[2]).execute("select name, start, end from kernels order by start").fetchall()
    rows = [r for r in rows if 'k_gap_' in r[0]]
    k = 0
    print(f"{'MB':>5} {'stores':>12} {'A us':>8} {'gap A->B us':>12} {'gap B->A us':>12}")
    for mb in SIZES_MB:
        for mode, name in MODES.items():
            chunk = rows[k:k + 2 * REP]
            k += 2 * REP
            a = [(chunk[2 * i][2] - chunk[2 * i][1]) / 1e3 for i in range(REP)]
            g = [(chunk[2 * i + 1][1] - chunk[2 * i][2]) / 1e3 for i in range(REP)]
            g2 = [(chunk[2 * i + 2][1] - chunk[2 * i + 1][2]) / 1e3 for i in range(REP - 1)]
            print(f"{mb:5d} {name:>12} {statistics.median(a):8.1f} {statistics.median(g):12.2f} {statistics.median(g2):12.2f}")
    chunk = rows[k:k + 2 * REP]
    g = [(chunk[2 * i + 1][1] - chunk[2 * i][2]) / 1e3 for i in range(REP)]
    g2 = [(chunk[2 * i + 2][1] - chunk[2 * i + 1][2]) / 1e3 for i in range(REP - 1)]
    print(f"  128 MB plain, a timing event recorded after every pair: gap A->B {statistics.median(g):.2f} us, gap B->[event]->A {statistics.median(g2):.2f} us")
