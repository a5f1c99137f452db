"""CPU pin of the lane-mask logic of the device sorting networks (line3dpp_amd/csrc/k_lists.hip: lanes_bit_clear,
cmp_exchange, list_sort_regs; k_match.hip: reg_sort uses the same network with per-lane direction flags).

On the GPU a compare-exchange step takes the comparison as a lane mask and combines it on the scalar unit with a
CONSTANT 64-bit pattern that says which lanes keep the smaller key of their pair.  This file restates those patterns
and the three kinds of stages (partner in the thread, in the wave, in another wave) for 64-lane waves in plain Python
and checks that every layout the kernels instantiate sorts random keys -- so a wrong pattern or a wrong `up` rule is
caught without a GPU."""
import random

import pytest

M = (1 << 64) - 1


def lanes_bit_clear(d):   # bit t set iff (t & d) == 0; all lanes for d >= 64
    return {1: 0x5555555555555555, 2: 0x3333333333333333, 4: 0x0F0F0F0F0F0F0F0F, 8: 0x00FF00FF00FF00FF,
            16: 0x0000FFFF0000FFFF, 32: 0x00000000FFFFFFFF}.get(d, M)


def cmp_exchange(v, o, keep_min):   # 64 lanes; keep_min: lanes that end up with the smaller key
    own_less = sum(1 << l for l in range(64) if v[l] < o[l])
    sel = ~(own_less ^ keep_min) & M
    return [v[l] if (sel >> l) & 1 else o[l] for l in range(64)]


def test_lane_patterns_are_what_they_claim():
    for d in (1, 2, 4, 8, 16, 32, 64):
        assert lanes_bit_clear(d) == sum(1 << t for t in range(64) if (t & d) == 0)


def lane_sort(keys):   # one key per lane (k_lists.hip, lists of up to 64 hypotheses)
    k = 2
    while k <= 64:
        j = k >> 1
        while j:
            keys = cmp_exchange(keys, [keys[l ^ j] for l in range(64)], ~(lanes_bit_clear(j) ^ lanes_bit_clear(k)) & M)
            j >>= 1
        k <<= 1
    return keys


def reg_sort(keys, KPT, WPL):   # thread t owns elements t*KPT .. (list_sort_regs<WPL, KPT>)
    GS = 64 * WPL
    N = GS * KPT
    v = [[keys[t * KPT + r] for r in range(KPT)] for t in range(GS)]

    def up_mask(k, r, wave_t):
        if k < KPT:
            return M if (r & k) == 0 else 0
        kk = k // KPT
        return lanes_bit_clear(kk) if kk < 64 else (M if (wave_t & kk) == 0 else 0)

    def stage(partner_of, keep_of):
        nonlocal v
        new = [row[:] for row in v]
        for w in range(WPL):
            wt = w * 64
            for r in range(KPT):
                own = [v[wt + l][r] for l in range(64)]
                oth = [v[partner_of(wt + l)][r] for l in range(64)]
                res = cmp_exchange(own, oth, keep_of(r, wt))
                for l in range(64):
                    new[wt + l][r] = res[l]
        v = new

    k = 2
    while k <= N:
        j = k >> 1
        while j >= 64 * KPT:                                        # partner in another wave
            d = j // KPT
            stage(lambda t: t ^ d, lambda r, wt: ~((M if (wt & d) == 0 else 0) ^ up_mask(k, r, wt)) & M)
            j >>= 1
        j = min(k >> 1, 32 * KPT)
        while j >= KPT:                                             # partner in this wave
            d = j // KPT
            stage(lambda t: (t & ~63) | ((t & 63) ^ d), lambda r, wt: ~(lanes_bit_clear(d) ^ up_mask(k, r, wt)) & M)
            j >>= 1
        jj = KPT // 2
        while jj:                                                   # partner in this thread
            if jj < k:
                for w in range(WPL):
                    wt = w * 64
                    for r in range(KPT):
                        if (r & jj) == 0:
                            a = [v[wt + l][r] for l in range(64)]
                            b = [v[wt + l][r | jj] for l in range(64)]
                            gt = sum(1 << l for l in range(64) if a[l] > b[l])
                            sw = ~(gt ^ up_mask(k, r, wt)) & M
                            for l in range(64):
                                if (sw >> l) & 1:
                                    v[wt + l][r], v[wt + l][r | jj] = b[l], a[l]
            jj >>= 1
        k <<= 1
    return [v[t][r] for t in range(GS) for r in range(KPT)]


def test_one_key_per_lane_network_sorts():
    rng = random.Random(5)
    for _ in range(10):
        ks = rng.sample(range(10 ** 6), 64)
        assert lane_sort(ks) == sorted(ks)


@pytest.mark.parametrize("KPT,WPL", [(2, 1), (4, 1), (2, 2), (4, 2), (2, 4), (4, 4)])
def test_register_networks_sort_for_every_instantiated_layout(KPT, WPL):
    rng = random.Random(KPT * 10 + WPL)
    for _ in range(3):
        ks = rng.sample(range(10 ** 7), 64 * WPL * KPT)
        assert reg_sort(ks, KPT, WPL) == sorted(ks)
