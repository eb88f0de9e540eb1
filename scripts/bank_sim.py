# LDS bank-conflict simulator for the halo2 kernel's pixel-fragment reads (ds_read_b128, 32x32x16 B operand)
import itertools, sys
GROUPS = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]

def conflicts(W, H, R, P, f, rowbytes=128):
    """extra LDS cycles per ds_read_b128 (sum over the 2 groups of one lane-half), averaged over blocks/taps"""
    npx = R * W
    nblk = (npx + 31) // 32
    tot = 0; n = 0; worst = 0
    slots_per_row = rowbytes // 16
    rows_per_win = 256 // rowbytes
    for blk in range(nblk):
        for ky in range(3):
            for kx in range(3):
                for s in range(rowbytes // 32):      # k16 step
                    for h in range(2):
                        col = 2 * s + h
                        for G in GROUPS:
                            seen = {}
                            for q in G:
                                pix = blk * 32 + q
                                if pix >= npx: pix = 0
                                y, x = divmod(pix, W)
                                r = (y + ky) * P + x + kx
                                c = col ^ f(r, r // P, P, W)
                                slot = (r % rows_per_win) * slots_per_row + c
                                addr = r * rowbytes + c * 16
                                seen.setdefault(slot, set()).add(addr)
                            extra = max(len(v) for v in seen.values()) - 1
                            tot += extra; n += 1; worst = max(worst, extra)
    return tot / n, worst

if __name__ == "__main__":
    cands = {
        "row&7": lambda r, yy, P, W: r & 7,
        "row&7 ^ bit4": lambda r, yy, P, W: (r & 7) ^ ((r >> 4) & 1),
        "(r>>1)&7": lambda r, yy, P, W: (r >> 1) & 7,
        "((r>>1)-yy*(J/2))&7": lambda r, yy, P, W: ((r >> 1) - yy * ((P - W) // 2)) & 7,
    }
    for (W, H, R) in [(38, 38, 10), (19, 19, 19), (76, 76, 5), (19, 19, 10), (38,38,5), (152,152,2)]:
        for P in sorted({W + 2, (W + 2 + 7) // 8 * 8}):
            for name, f in cands.items():
                if (P - W) % 2 and "J/2" in name: continue
                a, w = conflicts(W, H, R, P, f)
                print(f"W={W} R={R} P={P} {name:24s} avg extra cycles/group {a:.3f} worst {w}")
