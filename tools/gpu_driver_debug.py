import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_driver as td
tmp = "/tmp/drvdbg"; os.makedirs(tmp, exist_ok=True)
reads = os.path.join(tmp, "reads.fastq")
recs, truth = td.simulate(reads, n_reads=16, genome_len=2400, read_len=1500, err=0.10, seed=9)
env = dict(os.environ, PYTHONPATH=ROOT)
paf = os.path.join(tmp, "o.paf")
subprocess.check_call([sys.executable, os.path.join(ROOT, "tests/stubs/stub_overlapper.py"), reads, reads, paf])
print(open(paf).read().split("\n")[:3])
for flags, tag in ((["-f", "-p"], "hap"), (["-f"], "lin"), (["-f", "-p", "-u"], "hap-u")):
    out = subprocess.run([sys.executable, "-m", "vechat_amd.polish"] + flags + [reads, paf, reads], env=env, capture_output=True, text=True)
    lines = out.stdout.split("\n")
    got = {lines[i][1:].split()[0].rstrip("r"): lines[i + 1].encode() for i in range(0, len(lines) - 1, 2)}
    print("   header sample:", lines[0])
    print(tag, out.stderr.strip().split("\n")[-1])
    for n, s in recs[:6]:
        g = got.get(n)
        print("  ", n, "raw", len(s), "truth", len(truth[n]), "d_raw", td._fit_distance(s, truth[n]), "out", None if g is None else (len(g), td._fit_distance(g, truth[n])))
