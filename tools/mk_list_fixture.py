import os, sys, wave, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tests import flac_encode as FE
from wav2letter_amd import recipes
d = sys.argv[1]
os.makedirs(d + "/arch", exist_ok=True); os.makedirs(d + "/audio", exist_ok=True)
open(d + "/arch/net.arch", "w").write(recipes.conv_glu_small_arch(widths=(32, 48), kws=(5, 5)))
letters = ["|", "'"] + [chr(c) for c in range(ord("a"), ord("z") + 1)]
open(d + "/tokens.txt", "w").write("\n".join(letters) + "\n")
words = ["hello", "aaa", "bee", "zoo", "add"]
open(d + "/lexicon.txt", "w").write("".join(f"{w}\t{' '.join(w)} |\n" for w in words))
rng = np.random.default_rng(0)
lines = []
for k, (n, tr) in enumerate([(9600, "hello bee"), (6400, "aaa"), (8000, "zoo hello"), (4800, "bee"), (7300, "add zoo"), (5100, "hello")]):
    t = np.arange(n) / 16000.0
    sig = np.round((0.3 * np.sin(2 * np.pi * (200 + 150 * k) * t) + 0.05 * rng.normal(size=n)) * 30000).astype(np.int16)
    if k % 2:
        p = f"{d}/audio/u{k}.flac"; open(p, "wb").write(FE.encode(sig.astype(np.int64), kind="fixed", order=2, porder=2))
    else:
        p = f"{d}/audio/u{k}.wav"
        with wave.open(p, "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(sig.astype("<i2").tobytes())
    lines.append(f"u{k} {p} {n / 16.0:.1f} {tr}")
open(d + "/train.lst", "w").write("\n".join(lines) + "\n")
