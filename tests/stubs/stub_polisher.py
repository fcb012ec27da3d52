"""Test stand-in for the polisher command (tests/test_driver.py, no GPU): appends its argument vector to $VC_STUB_LOG and
writes the target sequences back as FASTA, names kept -- enough to follow the driver's file hand-off."""
import gzip
import json
import os
import sys

args = sys.argv[1:]
with open(os.environ["VC_STUB_LOG"], "a") as f:
    f.write(json.dumps(args) + "\n")
targets = args[-1]
op = gzip.open if targets.endswith(".gz") else open
with op(targets, "rt") as f:
    lines = [l.rstrip("\n") for l in f]
per = 4 if lines and lines[0].startswith("@") else 2
tag = os.environ.get("VC_STUB_TAG", "")          # "r": the tag fragment correction appends to a name (src/polisher.cpp:525)
for i in range(0, len(lines) - 1, per):
    sys.stdout.write(">" + lines[i][1:].split()[0] + tag + "\n" + lines[i + 1].upper() + "\n")
