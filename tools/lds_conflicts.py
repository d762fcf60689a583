"""LDS bank-conflict model of gfx950 (guide: MI355X_MICROARCH.md, "LDS"): a wave64 access is served in fixed lane groups, one
LDS cycle per group; every further distinct address on a busy bank inside a group adds a cycle.  The access patterns of the
fused encoder kernels are written down here as lane -> byte address functions, so a layout can be checked before it is built
(the round-2/3 swizzles were designed for lane groups {0-15}, {16-31}, ... which is NOT how ds_read_b128 groups its lanes).

    python tools/lds_conflicts.py            # extra cycles per access pattern, old and new layouts
"""
G128 = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
G128 = G128 + [[l + 32 for l in g] for g in G128]
GROUPS = {"read_b128": (G128, 16, 64), "read_b64": ([list(range(32)), list(range(32, 64))], 8, 64),
          "read_b64_tr": ([list(range(32)), list(range(32, 64))], 8, 64),
          "write_b64": ([list(range(16 * i, 16 * i + 16)) for i in range(4)], 8, 32),
          "write_b32": ([list(range(32)), list(range(32, 64))], 4, 32),
          "write_b128": ([list(range(8 * i, 8 * i + 8)) for i in range(8)], 16, 32)}


def cycles(kind, addr):
    """(base cycles, extra conflict cycles) of one wave instruction; addr(lane) -> byte address"""
    groups, nbytes, nbanks = GROUPS[kind]
    extra = 0
    for grp in groups:
        per_bank = {}
        for l in grp:
            a = addr(l)
            for d in range(nbytes // 4):
                per_bank.setdefault(((a // 4) + d) % nbanks, set()).add((a // 4) + d)
        extra += max(len(v) for v in per_bank.values()) - 1
    return len(groups), extra


def report(title, accesses):
    base = extra = 0
    print(title)
    for name, kind, count, fn in accesses:
        b, e = cycles(kind, fn)
        base += b * count
        extra += e * count
        print(f"    {name:44s} {kind:11s} x{count:3d}: {b} cycles + {e} conflict")
    print(f"    total per chunk and wave: {base} + {extra}  (conflict / (base + conflict) = {extra / (base + extra):.3f}; DMA writes not counted)")


def mlp_hs_bwd(f, ht=1, th=1, sp=0, w=5):
    """per chunk and wave; f = swizzle function of (row >> 2) & 3 for the 64-B-row tiles (S3, u, dz)"""
    g = lambda l: l >> 4
    j = lambda l: l & 15
    tok = lambda l: th * 16 + j(l)
    acc = []
    acc.append(("S1 / S2 A fragments (256-B rows)", "read_b128", 8, lambda l: (ht * 16 + j(l)) * 256 + (((0 * 4 + g(l)) ^ j(l)) << 4)))
    acc.append(("u / dz stage writes (64-B rows)", "write_b64", 4,
                lambda l: tok(l) * 64 + (((ht * 2 + (g(l) >> 1)) ^ f((tok(l) >> 2) & 3)) << 4) + (g(l) & 1) * 8))
    acc.append(("S3 A fragments (64-B rows)", "read_b128", 4,
                lambda l: ((ht * 4 + 1) * 16 + j(l)) * 64 + ((g(l) ^ f((((ht * 4 + 1) * 16 + j(l)) >> 2) & 3)) << 4)))
    acc.append(("dz B fragments (64-B rows)", "read_b128", 2, lambda l: tok(l) * 64 + ((g(l) ^ f((tok(l) >> 2) & 3)) << 4)))
    srow = lambda l: w * 16 + (l >> 2)
    acc.append(("u / dz rows for the global stores", "read_b128", 2, lambda l: srow(l) * 64 + (((l & 3) ^ f((srow(l) >> 2) & 3)) << 4)))
    return acc


def mlp_hs_fwd(hg=2, th=1):
    g = lambda l: l >> 4
    j = lambda l: l & 15
    tok = lambda l: th * 16 + j(l)
    acc = [("W1 / W2t A fragments", "read_b128", 16, lambda l: (hg * 32 + j(l)) * 256 + (((4 + g(l)) ^ j(l)) << 4)),
           ("u B fragments", "read_b128", 16, lambda l: tok(l) * 256 + (((8 + g(l)) ^ (tok(l) & 15)) << 4)),
           ("u writes (4 hidden of one token per lane)", "write_b64", 8,
            lambda l: tok(l) * 256 + ((((hg * 32 + 16 + 4 * g(l)) >> 3) ^ (tok(l) & 15)) << 4) + ((hg * 32 + 16 + 4 * g(l)) & 7) * 2)]
    return acc


def attn_block_fwd(xs_swizzled, w=2):
    """per wave (head dim 16); the partial-sum input form (every layer but the first)"""
    kh = lambda l: l >> 5
    l31 = lambda l: l & 31
    sw = lambda l: l & 15
    acc = [("a1 / Wqkv / Wo / o fragments (256-B rows)", "read_b128", 48, lambda l: (w * 32 + l31(l)) * 256 + (((3 * 2 + kh(l)) ^ sw(l)) << 4)),
           ("q / k fragments per head", "read_b128", 4, lambda l: l31(l) * 256 + ((((w * 32 + 16 + kh(l) * 8) >> 3) ^ sw(l)) << 4)),
           ("v^T gathers (72-B rows)", "read_b64", 8, lambda l: (w * 32 + (l31(l) % 16)) * 72 + (16 + 4 * kh(l)) * 2),
           ("a1 tile writes (2 features per lane)", "write_b32", 8, lambda l: 5 * 256 + (((l >> 2) ^ 5) << 4) + (l & 3) * 4),
           ("q / k / o writes (4 features of a token)", "write_b64", 12,
            lambda l: l31(l) * 256 + ((((w * 32 + 4 * kh(l) + 8) >> 3) ^ sw(l)) << 4) + ((w * 32 + 4 * kh(l) + 8) & 7) * 2),
           ("v^T writes (72-B rows)", "write_b64", 4, lambda l: (w * 32 + l31(l)) * 72 + (4 * kh(l) + 8) * 2)]
    if xs_swizzled:
        acc += [("combined input rows, fp32 (writes)", "write_b64", 8, lambda l: 5 * 512 + (((l >> 1) ^ 5) << 4) + (l & 1) * 8),
                ("residual rows, fp32 (epilogue reads)", "read_b128", 4, lambda l: l31(l) * 512 + (((w * 8 + kh(l) + 2) ^ sw(l)) << 4))]
    else:
        acc += [("combined input rows, fp32 (writes)", "write_b64", 8, lambda l: (5 * 128 + l * 2) * 4),
                ("residual rows, fp32 (epilogue reads)", "read_b128", 4, lambda l: (l31(l) * 128 + w * 32 + 4 * kh(l) + 8) * 4)]
    return acc


def attn_block_bwd(swz, w=1, hh=1):
    """per wave (head dim 16, two heads per wave)"""
    kh = lambda l: l >> 5
    l31 = lambda l: l & 31
    sw = lambda l: swz(l & 15)
    f0 = w * 32 + hh * 16

    def tr(l, ks2=1, second=0):
        gg, ig = l >> 4, l & 15
        fcol = f0 + 16 * (gg & 1) + 4 * (ig & 3)
        r = 16 * ks2 + 4 * kh(l) + (ig >> 2) + 8 * second
        return r * 256 + (((fcol >> 3) ^ swz(r & 15)) << 4) + (fcol & 7) * 2
    return [("Wo / dh / Wqkv / dqkv fragments (256-B rows)", "read_b128", 64, lambda l: (w * 32 + l31(l)) * 256 + (((5 * 2 + kh(l)) ^ sw(l)) << 4)),
            ("q / k / v / dO fragments per head", "read_b128", 8, lambda l: l31(l) * 256 + ((((f0 + kh(l) * 8) >> 3) ^ sw(l)) << 4)),
            ("transposing reads of k, q, dO (first rows)", "read_b64_tr", 12, lambda l: tr(l)),
            ("transposing reads of k, q, dO (rows + 8)", "read_b64_tr", 12, lambda l: tr(l, second=1)),
            ("dO / dq / dk / dv writes (4 features of a token)", "write_b64", 16,
             lambda l: l31(l) * 256 + ((((f0 + 4 * kh(l)) >> 3) ^ sw(l)) << 4) + ((f0 + 4 * kh(l)) & 7) * 2)]


if __name__ == "__main__":
    report("mlp_hs_bwd, round-3 layout (piece ^ ((row >> 2) & 3))", mlp_hs_bwd(lambda x: x))
    report("mlp_hs_bwd, round-4 layout (piece ^ perm[(row >> 2) & 3], perm = 0 2 3 1)", mlp_hs_bwd(lambda x: (0x78 >> (2 * x)) & 3))
    report("mlp_hs_fwd", mlp_hs_fwd())
    report("attn_block_fwd, round-3 layout (linear fp32 input rows)", attn_block_fwd(False))
    report("attn_block_fwd, round-4 layout (fp32 input rows slot-swizzled)", attn_block_fwd(True))
    report("attn_block_bwd, round-3 layout (slot ^ (row & 15))", attn_block_bwd(lambda r: r))
    report("attn_block_bwd, round-4 layout (slot ^ bit-pair-swapped row)", attn_block_bwd(lambda r: ((r & 3) << 2) | ((r >> 2) & 3)))
