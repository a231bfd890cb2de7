"""Synthetic back-off n-gram model in ARPA text (measurement input, SURVEY.md
section 8d "Synthetic 4-gram LM"): vocabulary w0..w{W-1} (or t0..t{N-1} for a
token LM) + <s>, </s>, <unk>; every unigram; random higher-order n-grams closed
under prefix and suffix; log10 probabilities / back-offs = -(24-bit int) * 2^-20
(exact in float32 and exactly round-trippable through decimal text)."""
import numpy as np


class _Sm64:
    def __init__(self, seed):
        self.s = seed & ((1 << 64) - 1)

    def next(self):
        M = (1 << 64) - 1
        self.s = (self.s + 0x9E3779B97F4A7C15) & M
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        return z ^ (z >> 31)

    def val(self):
        return -float(np.float32((self.next() >> 40) * 2.0 ** -20))


def words(n, prefix="w"):
    return ["%s%d" % (prefix, i) for i in range(n)]


def write_arpa(path, vocab, order=4, counts=(0, 2000, 1000, 500), seed=99):
    """vocab: list of user words.  counts[k] = number of random (k+1)-grams to
    draw for k >= 1 (closure adds more).  Returns the number of n-grams."""
    g = _Sm64(seed)
    lm_words = ["<unk>", "<s>", "</s>"] + list(vocab)
    V = len(lm_words)
    grams = [dict() for _ in range(order)]
    for w in range(V):
        grams[0][(w,)] = (g.val(), g.val())

    def ensure(t):
        k = len(t) - 1
        if t in grams[k]:
            return
        grams[k][t] = (g.val(), g.val())
        if k > 0:
            ensure(t[:-1])  # prefix (holds the back-off of the context)
            ensure(t[1:])   # suffix (what scoring backs off to)

    for k in range(1, order):
        for _ in range(counts[k] if k < len(counts) else 0):
            t = tuple(int(g.next() % V) for _ in range(k + 1))
            if 2 in t[:-1] or 1 in t[1:]:
                continue  # </s> never inside a context, <s> only first
            ensure(t)
    with open(path, "w") as f:
        f.write("\\data\\\n")
        for k in range(order):
            f.write("ngram %d=%d\n" % (k + 1, len(grams[k])))
        for k in range(order):
            f.write("\n\\%d-grams:\n" % (k + 1))
            for t, (p, b) in sorted(grams[k].items()):
                ws = " ".join(lm_words[i] for i in t)
                if k + 1 < order:
                    f.write("%.9g\t%s\t%.9g\n" % (p, ws, b))
                else:
                    f.write("%.9g\t%s\n" % (p, ws))
        f.write("\n\\end\\\n")
    return sum(len(x) for x in grams)
