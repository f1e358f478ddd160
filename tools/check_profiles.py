#!/usr/bin/env python3
"""tools/check_profiles.py rNN -- are the round's committed profiles of the SHIPPED library?
Every profiles/rNN_* file that carries a `library_sha256` (written on the GPU box by
tools/profile_round.sh, profile_large.sh, pmc_counters.sh) must name the library now in the
tree; tools/final_check.sh refuses to finish otherwise (round 5's
counters were of the build before its last kernel change).
(The build is reproducible where it stands: `make` in a clean checkout at the same tree path --
/root/repo -- gives the same bytes, hence the same hash; at another path hipcc's per-file unit
ids, which are made from the source paths, differ in ~600 bytes of symbol names.)"""
import glob, hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r06"
lib = os.path.join(ROOT, "kaldi-decoder_amd", "python", "kaldi_decoder", "lib", "libkd_hip.so")
want = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16]
bad = []
seen = 0
for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"{rnd}_*.json"))):
    try:
        h = json.load(open(f)).get("library_sha256")
    except ValueError:
        continue
    if h is None:
        continue
    seen += 1
    print("%-44s %s %s" % (os.path.basename(f), h, "ok" if h == want else "STALE (shipped: %s)" % want))
    if h != want:
        bad.append(f)
if bad or not seen:
    print("profiles of another build than the shipped library" if bad else "no profile of this round names a library")
    sys.exit(1)
