"""Synthetic attack-string corpora, generated on the device.

The string generator is the reference's own `pumped_string` (matchers/example_runner.cpp:15-29,
identical to matcher.py:26-38): only pump parts 0 and 1 are used, the result is
    prefix + (res + pump[1]) * del_count + res + suffix,   res = pump[0] repeated
with the non-accumulating prefix of matcher.py:58.  pump.txt of every example is restated in
EXAMPLES (regex from test/example_N/regexp.txt line 1 / README.md:81-92).
"""
import math

import numpy as np

# example -> (regex, pump parts, suffix, prefix)
EXAMPLES = {
    1: ("({a*}:1&1)*", ["a"], "b", ""),
    2: ("{(a|bb)*}:1aaba(&1|bb*aa)*", ["bbaa", "aaba", "bbaa"], "c", ""),
    3: ("{{a*}:1(&1)*}:2b&2a*", ["a", "b", "a"], "aab", ""),
    4: ("({a*}:1&1a*)*", ["a"], "b", ""),
    5: ("{a*}:1c{&1}:2c(&1|&2)*", ["aa"], "b", "aacaac"),
    6: ("({a*}:1b|&1)*", ["a"], "c", "aaab"),
    7: ("({a*}:1)*b&1", ["a", "b", "a"], "b", ""),
    8: ("(({a*}:1|b)(&1|b))*", ["a", "b", "a"], "c", "bb"),
    9: ("(({aa*b}:1(&1)*)|b(b|a*)*)*", ["bbaaa"], "c", ""),
    10: ("({a*}:1b|b&1)*c&1", ["aababba"], "cab", ""),
}

# the reference's further examples (test/example_11..17: regexp.txt, pump.txt; no README row): not part of the benchmark corpus,
# used by the tests
MORE_EXAMPLES = {
    11: ("({a*}:1b&1b)*", ["aa", "b", "aa"], "bbc", ""),
    12: ("(({a*}:1b&1b)*)*", ["aa", "b", "aa"], "bbc", ""),
    13: ("ba{aa*}:1a&1*", ["a", "a", "a"], "b", "ba"),
    14: ("{(a*|b*)}:1b(&1|b)*", ["b", "b", "b", "b", "b"], "bc", ""),
    15: ("{a*}:1c{a*}:2c(&1|&2)*", ["aa"], "b", "aacaac"),
    16: ("({a*}:1&1|(a*|b)a)*", ["baaaa"], "b", ""),
    17: ("(&1{a*}:1|(a*|b)a)*", ["baaaa"], "b", ""),
}
ALL_EXAMPLES = {**EXAMPLES, **MORE_EXAMPLES}


def pumped_string(n, pump):
    """example_runner.cpp:15-29, host version (tests, small samples)."""
    pump_count = len(pump) // 2 + 1
    del_count = len(pump) - pump_count
    res = pump[0]
    while len(res) + len(pump[0]) < (n - del_count) // pump_count:
        res += pump[0]
    out = ""
    for _ in range(del_count):
        out += res + pump[1]
    return out + res


def _repeats(n, pump):
    """number of copies of pump[0] in `res` for pump size n (vectorised over n)."""
    u = len(pump[0])
    pump_count = len(pump) // 2 + 1
    del_count = len(pump) - pump_count
    limit = (n - del_count) // pump_count
    # smallest m >= 1 with m*u + u >= limit
    m = np.maximum(1, -(-(limit - u) // u))
    return m.astype(np.int64), del_count


def pump_sizes(n_strings, seed, lo=1024, hi=65536):
    """pump sizes n, log-uniform in [lo, hi]."""
    rng = np.random.Generator(np.random.Philox(seed))
    return np.exp(rng.uniform(math.log(lo), math.log(hi), size=n_strings)).astype(np.int64)


def layout(example, sizes, with_suffix):
    """per-string lengths for `example` given pump sizes and suffix flags (numpy)."""
    regex, pump, suffix, prefix = EXAMPLES[example]
    m, dc = _repeats(sizes, pump)
    u = len(pump[0])
    s = len(pump[1]) if len(pump) > 1 else 0
    res_len = m * u
    block = res_len + s
    pumped_len = dc * block + res_len
    lens = len(prefix) + pumped_len + np.where(with_suffix, len(suffix), 0)
    return {"res_len": res_len, "block": block, "pumped_len": pumped_len, "lens": lens.astype(np.int64), "dc": dc}


def host_strings(example, sizes, with_suffix):
    regex, pump, suffix, prefix = ALL_EXAMPLES[example]
    return [(prefix + pumped_string(int(n), pump) + (suffix if w else "")).encode() for n, w in zip(sizes, with_suffix)]


def device_batch(example, sizes, with_suffix, device, chunk_bytes=1 << 27):
    """Build the batch for `example` on `device`: returns (bytes uint8 [total + 64], offsets int64 [n+1])."""
    import torch
    regex, pump, suffix, prefix = EXAMPLES[example]
    lay = layout(example, sizes, with_suffix)
    lens = lay["lens"]
    off = np.zeros(len(lens) + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    total = int(off[-1])
    out = torch.zeros(total + 64, dtype=torch.uint8, device=device)
    d_off = torch.from_numpy(off).to(device)

    def tab(s):
        return torch.tensor(list((s or "\0").encode()), dtype=torch.uint8, device=device)

    t_unit, t_sep, t_pre, t_suf = tab(pump[0]), tab(pump[1] if len(pump) > 1 else ""), tab(prefix), tab(suffix)
    u, P = len(pump[0]), len(prefix)
    d_res = torch.from_numpy(lay["res_len"]).to(device)
    d_block = torch.from_numpy(lay["block"]).to(device)
    d_pl = torch.from_numpy(lay["pumped_len"]).to(device)
    a = 0
    n = len(lens)
    while a < n:
        b = int(np.searchsorted(off, off[a] + chunk_bytes, side="right")) - 1
        b = max(b, a + 1)
        b = min(b, n)
        lo, hi = int(off[a]), int(off[b])
        cnt = hi - lo
        sid = torch.repeat_interleave(torch.arange(a, b, device=device), d_off[a + 1:b + 1] - d_off[a:b], output_size=cnt)
        j = torch.arange(lo, hi, device=device) - d_off[sid]
        j2 = j - P
        block = d_block[sid]
        r = torch.where(j2 >= 0, j2 % block, torch.zeros_like(j2))
        res_len = d_res[sid]
        in_res = r < res_len
        c_unit = t_unit[(r % u).clamp_(0, len(t_unit) - 1)]
        c_sep = t_sep[(r - res_len).clamp_(0, len(t_sep) - 1)]
        c = torch.where(in_res, c_unit, c_sep)
        after = j2 - d_pl[sid]
        c = torch.where(after >= 0, t_suf[after.clamp(0, len(t_suf) - 1)], c)
        c = torch.where(j2 < 0, t_pre[j.clamp(0, len(t_pre) - 1)], c)
        out[lo:hi] = c
        del sid, j, j2, block, r, res_len, in_res, c_unit, c_sep, c, after
        a = b
    return out, d_off
