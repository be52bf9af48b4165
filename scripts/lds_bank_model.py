"""LDS bank-conflict model of gfx950 (MI355X_MICROARCH.md, LDS table) for the access sites of the workgroup-tiled MLP kernels
(csrc/mlp_kernels.hip).  For a wave instruction: lanes are served in fixed groups, one LDS cycle per group when conflict-free;
each extra distinct dword on a busy bank within a group adds a cycle.  Prints cycles per wave instruction and per site, for the
layout parameters given on the command line -- used to choose the tile paddings / swizzles (profiles/r02_exp_notes.txt).

    python scripts/lds_bank_model.py            # current layouts, WIDTH 128 and 64
"""
import itertools
import sys

G128 = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
        list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)), list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
C16 = [list(range(16 * i, 16 * i + 16)) for i in range(4)]
C8 = [list(range(8 * i, 8 * i + 8)) for i in range(8)]
H32 = [list(range(0, 32)), list(range(32, 64))]
# (lane groups, number of banks) per instruction
KIND = {"read_b32": (H32, 32), "read_b64": (H32, 64), "read_b128": (G128, 64), "write_b16": (H32, 32), "write_b32": (H32, 32), "write_b64": (C16, 32),
        "write_b128": (C8, 32)}
WIDTH_BYTES = {"read_b32": 4, "read_b64": 8, "read_b128": 16, "write_b16": 2, "write_b32": 4, "write_b64": 8, "write_b128": 16}


def cycles(kind, addr_of_lane, active=None):
    groups, n_banks = KIND[kind]
    nbytes = WIDTH_BYTES[kind]
    total = 0
    for grp in groups:
        per_bank = {}
        for lane in grp:
            if active is not None and not active(lane):
                continue
            a = addr_of_lane(lane)
            for dw in range(a // 4, (a + max(nbytes, 1) - 1) // 4 + 1):
                per_bank.setdefault(dw % n_banks, set()).add(dw)
        total += max([len(v) for v in per_bank.values()], default=0)
    return total


def ideal(kind):
    return len(KIND[kind][0])


def report(name, kind, count, fn):
    """count: wave instructions of this site per tile and wave; fn(variant) -> addr_of_lane for a representative instance"""
    c = [cycles(kind, f) for f in fn]
    avg = sum(c) / len(c)
    print(f"  {name:58s} {kind:11s} x{count:5d}  {avg:6.1f} cyc/instr (ideal {ideal(kind)})  -> {avg * count:9.0f} cycles")
    return avg * count, ideal(kind) * count


def mlp_backward_sites(W, S, SP, LDW, swz):
    """Access sites of k_mlp_backward<W, HM> for one hidden matrix + the staging of one saved activation tile.
    swz(k, i) -> physical sample index of logical sample i in feature row k (feature-major tiles)."""
    NW, NT, NTP, NB = W // 16, S // 16, S // 32, W // 16
    lr = lambda l: l & 15
    g = lambda l: l >> 4
    tot = idl = 0

    def ft(k, i):  # byte address of element (feature k, sample i) in a feature-major tile
        return (k * SP + swz(k, i)) * 2

    # A. staging of one saved activation tile: thread c -> (i = c / (W/8), cc = c % (W/8)), 8 two-byte stores
    n_iter = S * W // 8 // (NW * 64)
    fns = [(lambda l, j=j, it=it: ft(8 * ((it * NW * 64 + l) % (W // 8)) + j, (it * NW * 64 + l) // (W // 8))) for j in range(8) for it in range(n_iter)]
    a, b = report("stage hT: 2-byte transposing stores", "write_b16", 8 * n_iter * NW, fns)
    tot, idl = tot + a, idl + b
    # C. weight gradient: B operand h4 reads of hj rows (16b+lr), samples 32tp + 4g (+16)
    fns = [(lambda l, bb=bb, tp=tp, o=o: ft(16 * bb + lr(l), 32 * tp + o + 4 * g(l))) for bb in range(NB) for tp in range(NTP) for o in (0, 16)]
    a, b = report("dW: h4 reads of A_j^T", "read_b64", 2 * NB * NTP * NW, fns)
    tot, idl = tot + a, idl + b
    # C. chain: A operand h8 reads of cur rows (16t+lr), columns 32kb + 8g
    fns = [(lambda l, t=t, kb=kb: ((16 * t + lr(l)) * LDW + 32 * kb + 8 * g(l)) * 2) for t in range(NT) for kb in range(W // 32)]
    a, b = report("chain: h8 reads of dA (sample-major)", "read_b128", NT * (W // 32) * NW, fns)
    tot, idl = tot + a, idl + b
    # C. mask reads h4 of hj row (16w+lr), samples 16t+4g
    fns = [(lambda l, w=w, t=t: ft(16 * w + lr(l), 16 * t + 4 * g(l))) for w in range(NW) for t in range(NT)]
    a, b = report("chain: h4 mask reads of A_j^T", "read_b64", NT * NW, fns)
    tot, idl = tot + a, idl + b
    # C. dA stores: 2-byte, (sample 16t+4g+r, neuron 16w+lr)
    fns = [(lambda l, w=w, t=t, r=r: ((16 * t + 4 * g(l) + r) * LDW + 16 * w + lr(l)) * 2) for w in range(NW) for t in range(NT) for r in range(4)]
    a, b = report("chain: 2-byte stores of dA (sample-major)", "write_b16", 4 * NT * NW, fns)
    tot, idl = tot + a, idl + b
    print(f"  total per hidden matrix and tile: {tot:.0f} LDS cycles (conflict-free: {idl})")
    return tot


if __name__ == "__main__":
    for W in (128, 64):
        S = 64
        print(f"k_mlp_backward<{W}>: S = {S}, SP = S + 8, LDW = W + 8, no swizzle")
        mlp_backward_sites(W, S, S + 8, W + 8, lambda k, i: i)
        print(f"k_mlp_backward<{W}>: sample blocks of 8 XOR-ed with (k >> 3) & 7")
        mlp_backward_sites(W, S, S + 8, W + 8, lambda k, i: i ^ (((k >> 3) & 7) << 3))


def search():
    """Paddings / swizzles that minimise the modelled LDS cycles of the sites above."""
    lr = lambda l: l & 15
    g = lambda l: l >> 4
    print("sample-major tile, b128 reads of rows 16t + lr at column 32kb + 8g, 2-byte stores (sample 16t+4g+r, neuron 16w+lr), b64 stores (row 16t+lr, col 16w+4g):")
    for W in (64, 128):
        for pad in range(0, 80, 8):
            LDW = W + pad
            rd = cycles("read_b128", lambda l: (lr(l) * LDW + 8 * g(l)) * 2)
            st = max(cycles("write_b16", lambda l, r=r: ((4 * g(l) + r) * LDW + lr(l)) * 2) for r in range(4))
            st64 = cycles("write_b64", lambda l: (lr(l) * LDW + 4 * g(l)) * 2)
            print(f"  W {W:3d} LDW = W + {pad:2d}: read_b128 {rd} (ideal 4), write_b16 {st} (ideal 2), write_b64 {st64} (ideal 4)")
    print("feature-major tile [k][SP], S = 64: staging stores (cc = lane % (W/8), i = lane / (W/8)), b64 reads rows 16b+lr at 32tp+4g, b128 reads rows at 32tp+8g:")
    for W in (64, 128):
        for pad in range(0, 72, 8):
            for mask in (0, 1, 3, 7):
                for shift in (3, 2):
                    SP = 64 + pad
                    swz = lambda k, i, mask=mask, shift=shift: i ^ (((k >> 3) & mask) << shift) if shift == 3 else i ^ ((((k >> 3) & mask) << 2) & 60)
                    ft = lambda k, i: (k * SP + swz(k, i)) * 2
                    n8 = W // 8
                    st = max(cycles("write_b16", lambda l, j=j: ft(8 * (l % n8) + j, l // n8)) for j in range(8))
                    r64 = max(cycles("read_b64", lambda l, o=o: ft(lr(l), o + 4 * g(l))) for o in (0, 16, 32, 48))
                    r64b = max(cycles("read_b64", lambda l, o=o: ft(16 + lr(l), o + 4 * g(l))) for o in (0, 16, 32, 48))
                    ok128 = shift == 3 or mask == 0
                    r128 = max(cycles("read_b128", lambda l, o=o: ft(lr(l), o + 8 * g(l))) for o in (0, 32)) if ok128 else -1
                    if st <= 8:
                        print(f"  W {W:3d} SP = 64 + {pad:2d}, xor ((k>>3)&{mask})<<{shift}: stage write_b16 {st} (ideal 2), read_b64 {max(r64, r64b)} (ideal 2), read_b128 {r128} (ideal 4)")


if len(sys.argv) > 1 and sys.argv[1] == "search":
    search()


def wide_sites(LDa, LDw, swz_a=lambda r, c: c, swz_w=lambda r, c: c, verbose=True):
    """Access sites of k_mlp_train_wide (mlp_train_wide.hip) per 32-sample tile and wave, HM = 3: sample-major activation / gradient
    tiles [32][LDa], weight images [128][LDw]; swz(row, col) -> physical column.  ds_read_b64_tr_b16 is modelled as read_b64."""
    lr = lambda l: l & 15
    g = lambda l: l >> 4
    i4 = lambda l: (l & 15) >> 2
    q4 = lambda l: (l & 15) & 3
    A = lambda r, c: (r * LDa + swz_a(r, c)) * 2
    W = lambda r, c: (r * LDw + swz_w(r, c)) * 2
    tot = idl = 0
    def rep(name, kind, count, fns):
        nonlocal tot, idl
        c = [cycles(kind, f) for f in fns]
        avg = sum(c) / len(c)
        if verbose:
            print(f"  {name:62s} {kind:10s} x{count:3d} {avg:5.1f} cyc (ideal {ideal(kind)})")
        tot += avg * count
        idl += ideal(kind) * count
    w_all, kb_all, t_all = range(8), range(4), range(2)
    rep("fwd: A operand rows of W (16w+lr, 32kb+8g)", "read_b128", 12, [(lambda l, w=w, kb=kb: W(16 * w + lr(l), 32 * kb + 8 * g(l))) for w in w_all for kb in kb_all])
    rep("fwd / chain: B or A operand rows of a tile (16t+lr, 32kb+8g)", "read_b128", 48 + 2, [(lambda l, t=t, kb=kb: A(16 * t + lr(l), 32 * kb + 8 * g(l))) for t in t_all for kb in kb_all])
    rep("fwd: epilogue stores (16t+lr, 16w+4g)", "write_b64", 8, [(lambda l, t=t, w=w: A(16 * t + lr(l), 16 * w + 4 * g(l))) for t in t_all for w in w_all])
    rep("chain: B operand out of W, transposed (32kb+8g+4h+i/4, 16w+4(i%4))", "read_b64", 24, [(lambda l, w=w, kb=kb, h=h: W(32 * kb + 8 * g(l) + 4 * h + i4(l), 16 * w + 4 * q4(l))) for w in w_all for kb in kb_all for h in (0, 1)])
    rep("dW: B operand out of a tile, transposed (16h+4g+i/4, 16b+4(i%4))", "read_b64", 48, [(lambda l, b=b, h=h: A(16 * h + 4 * g(l) + i4(l), 16 * b + 4 * q4(l))) for b in range(8) for h in (0, 1)])
    rep("masks: (16t+4g+i/4, 16w+4(i%4))", "read_b64", 8, [(lambda l, t=t, w=w: A(16 * t + 4 * g(l) + i4(l), 16 * w + 4 * q4(l))) for t in t_all for w in w_all])
    rep("dA stores, 2 bytes (16t+4g+r, 16w+lr)", "write_b16", 32, [(lambda l, t=t, w=w, r=r: A(16 * t + 4 * g(l) + r, 16 * w + lr(l))) for t in t_all for w in w_all for r in range(4)])
    if verbose:
        print(f"  total {tot:.0f} LDS cycles per tile and wave (conflict-free {idl})")
    return tot


if len(sys.argv) > 1 and sys.argv[1] == "wide":
    print("k_mlp_train_wide, padded rows: LD = 136 for tiles and weights")
    wide_sites(136, 136)
    for lda, ldw in ((136, 144), (144, 136), (144, 144), (152, 152), (132, 132), (140, 140)):
        print(f"LDa {lda} LDw {ldw}: {wide_sites(lda, ldw, verbose=False):.0f}")
    fa = lambda r, c: c ^ ((r & 7) << 4)
    fw = lambda r, c: c ^ (((r & 3) | ((r >> 1) & 4)) << 4)
    print("unpadded rows (128), 16-byte chunks XOR-ed: tiles with (row & 7) << 1, weights with (bits 0, 1, 3 of row) << 1")
    wide_sites(128, 128, fa, fw)
