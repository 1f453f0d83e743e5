"""LDS bank model (MI355X_MICROARCH.md, LDS table) for k_threshold_mfma's tile accesses: which per-column rotation of the four
row octets of a tile column (80-byte pitch) makes the 16-byte reads of the column pass and of the preset product, and the
8-byte writes of the blur, conflict-free.  Brute force over rotations sigma(c mod 16) in 0..3 (hill climbing)."""
import itertools, random

GROUPS_B128 = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
               list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
               list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
               list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
GROUPS_W64 = [list(range(16 * g, 16 * g + 16)) for g in range(4)]
PITCH_DW = 20


def cycles(groups, addr_dw, width_dw, nbanks):
    """addr_dw[lane] = first dword; returns LDS cycles (sum over groups of the worst bank multiplicity)"""
    total = 0
    for g in groups:
        banks = {}
        for lane in g:
            for k in range(width_dw):
                a = addr_dw[lane] + k
                banks.setdefault(a % nbanks, set()).add(a)
        total += max(len(v) for v in banks.values())
    return total


def cost(sigma, verbose=False):
    tot = 0
    detail = []
    # column-pass read: lane (l16 = column c, q): octet o(q) = [h0, h0 + 1, h1, h1 + 1] with (h0, h1) = (0, 2) or (2, 0) by step parity
    for shift in (0, 8):                      # the preset product reads column l16 + 8
        for h0, h1 in ((0, 2), (2, 0)):
            addr = {}
            for lane in range(64):
                l16, q = lane & 15, lane >> 4
                c = l16 + shift
                o = (h0 if q < 2 else h1) + (q & 1)
                addr[lane] = PITCH_DW * c + 4 * ((o + sigma[c % 16]) % 4)
            cy = cycles(GROUPS_B128, addr, 4, 64)
            detail.append(("read", shift, h0, cy))
            tot += cy - 4
    # blur write: lane (l16 = column, q) writes rows 16 PAR + 4 q .. + 3: octet 2 PAR + (q >> 1), half (q & 1)
    for par in (0, 1):
        addr = {}
        for lane in range(64):
            l16, q = lane & 15, lane >> 4
            o = 2 * par + (q >> 1)
            addr[lane] = PITCH_DW * l16 + 4 * ((o + sigma[l16]) % 4) + 2 * (q & 1)
        cy = cycles(GROUPS_W64, addr, 2, 32)
        detail.append(("write", par, 0, cy))
        tot += max(0, cy - 6)                 # a store costs 6 cycles anyway (register transfer)
    if verbose:
        print(detail)
    return tot


def main():
    base = [0] * 16
    print("no rotation:", cost(base, True))
    random.seed(1)
    best, best_c = base, cost(base)
    for trial in range(200):
        s = [random.randrange(4) for _ in range(16)]
        c = cost(s)
        improved = True
        while improved:
            improved = False
            for i in range(16):
                for v in range(4):
                    t = list(s); t[i] = v
                    ct = cost(t)
                    if ct < c:
                        s, c, improved = t, ct, True
        if c < best_c:
            best, best_c = s, c
    print("best rotation:", best, "extra cycles", best_c)
    cost(best, True)


if __name__ == "__main__":
    main()
