#!/usr/bin/env python
"""Ties profile summaries to the build they were collected with:

    python3 tools/stamp_profiles.py <dir> <tag>

writes igi_build_info() -- the hash of the sources libigi_hip.so was compiled from -- into every <dir>/<tag>_*.json (top-level
key "build") and <dir>/<tag>_*.csv (first line "# build: <hash>"; readers skip '#' lines).  bench.py attaches the
profile-sourced roofline fields (traffic, pmc_bytes_per_step, frac_rocprof) only when that hash equals the loaded library's.
"""
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build_hash():
    import __graft_entry__ as g      # read from the library file's bytes: no GPU, no dlopen
    h = g.library_hash()
    if h is None:
        raise SystemExit("libigi_hip.so carries no source hash")
    return h


def stamp(path, h):
    if path.endswith(".json"):
        try:
            d = json.load(open(path))
        except ValueError:
            # a log with one JSON line at its end (bench output under rocprofv3): stamp that line
            lines = open(path).read().splitlines()
            for i in range(len(lines) - 1, -1, -1):
                if lines[i].startswith("{"):
                    d = json.loads(lines[i])
                    d["build"] = h
                    lines[i] = json.dumps(d)
                    open(path, "w").write("\n".join(lines) + "\n")
                    return True
            return False
        if not isinstance(d, dict):
            return False
        d["build"] = h
        json.dump(d, open(path, "w"), indent=1)
        return True
    if path.endswith(".csv"):
        body = open(path).read()
        if body.startswith("# build:"):
            body = body.split("\n", 1)[1]
        open(path, "w").write(f"# build: {h}\n" + body)
        return True
    return False


def read_build(path):
    """the build hash a profile summary was stamped with, or None"""
    try:
        if path.endswith(".csv"):
            first = open(path).readline()
            return first.split(":", 1)[1].strip() if first.startswith("# build:") else None
        d = json.load(open(path))
        return d.get("build") if isinstance(d, dict) else None
    except (OSError, ValueError):
        return None


def main():
    d, tag = sys.argv[1], sys.argv[2]
    h = build_hash()
    n = 0
    for p in sorted(glob.glob(os.path.join(d, f"{tag}_*.json")) + glob.glob(os.path.join(d, f"{tag}_*.csv"))):
        try:
            if stamp(p, h):
                n += 1
        except (ValueError, OSError) as e:     # a log that is not one JSON document: left as it is, and said so
            print(f"[stamp] skipped {os.path.basename(p)}: {type(e).__name__}")
    print(f"[stamp] {n} files under {d} stamped with build {h}")


if __name__ == "__main__":
    main()
