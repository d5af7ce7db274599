#!/usr/bin/env python3
"""Drops the hipfact shim into a chrhansk/sleqp checkout (v1.0.2) and registers the backend.

    python scripts/overlay_sleqp.py /path/to/sleqp [--aug-jac] [--tr] [--dry-run]

What it does (INTEGRATION.md sections 1, 2 and 6):
  * copies shim/fact_hipfact.{c,h} to src/main/fact/ and shim/SearchFactHIPFACT.cmake to cmake/
  * appends  add_fact(NAME "HIPFACT" SOURCES fact/fact_hipfact.c)  to cmake/SearchFact.cmake, right after
    the last add_fact(...) block (registration mechanism: cmake/SearchFact.cmake:11-83)
  * --aug-jac: copies shim/aug_jac_hipfact.{c,h} to src/main/aug_jac/, adds the source to the HIPFACT backend
    and patches create_aug_jac (trial_point.c:105-108) to create the device-assembly AugJac
  * --tr: copies shim/tr_hipfact.{c,h} to src/main/tr/ and adds the source to the backend
  * --mat: copies shim/mat_hipfact.{c,h} to src/main/sparse/ and adds the source to the backend
Then configure with  cmake -DSLEQP_FACT=HIPFACT -DHIPFACT_ROOT=<this repository> ...
The script is idempotent; --dry-run prints what would change and touches nothing.
"""
import argparse
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# create_aug_jac (trial_point.c:63-159) creates the standard AugJac in two places (AUTO without PSD, STANDARD)
TRIAL_POINT_OLD = re.compile(r"sleqp_standard_aug_jac_create\(\s*&solver->aug_jac,\s*problem,\s*settings,\s*solver->fact\)")
TRIAL_POINT_NEW = ("sleqp_hipfact_aug_jac_create(&solver->aug_jac, problem, settings, NULL /* hipfact: KKT assembled and "
                   "factorised on the device instead of the standard AugJac over solver->fact */)")


def plan(sleqp, aug_jac, tr, mat=False):
    copies = [("shim/fact_hipfact.c", "src/main/fact/fact_hipfact.c"),
              ("shim/fact_hipfact.h", "src/main/fact/fact_hipfact.h"),
              ("shim/SearchFactHIPFACT.cmake", "cmake/SearchFactHIPFACT.cmake")]
    sources = ["fact/fact_hipfact.c"]
    if aug_jac:
        copies += [("shim/aug_jac_hipfact.c", "src/main/aug_jac/aug_jac_hipfact.c"),
                   ("shim/aug_jac_hipfact.h", "src/main/aug_jac/aug_jac_hipfact.h")]
        sources.append("aug_jac/aug_jac_hipfact.c")
    if tr:
        copies += [("shim/tr_hipfact.c", "src/main/tr/tr_hipfact.c"), ("shim/tr_hipfact.h", "src/main/tr/tr_hipfact.h")]
        sources.append("tr/tr_hipfact.c")
    if mat:
        copies += [("shim/mat_hipfact.c", "src/main/sparse/mat_hipfact.c"), ("shim/mat_hipfact.h", "src/main/sparse/mat_hipfact.h")]
        sources.append("sparse/mat_hipfact.c")
    return copies, sources


def register(text, sources):
    """Returns cmake/SearchFact.cmake with the HIPFACT backend registered after the last add_fact block."""
    block = "add_fact(\n  NAME \"HIPFACT\"\n  SOURCES\n" + "".join(f"  {s}\n" for s in sources).rstrip("\n") + ")\n"
    text = re.sub(r"add_fact\(\s*NAME \"HIPFACT\".*?\)\n\n?", "", text, flags=re.S)  # idempotent: replace an old block
    last = None
    for m in re.finditer(r"add_fact\(.*?\)\n", text, flags=re.S):
        last = m
    if last is None:
        raise SystemExit("cmake/SearchFact.cmake: no add_fact(...) block found - is this a sleqp v1.0.2 checkout?")
    return text[: last.end()] + "\n" + block + text[last.end():]


def patch_trial_point(text):
    if "sleqp_hipfact_aug_jac_create" in text:
        return text
    text, count = TRIAL_POINT_OLD.subn(TRIAL_POINT_NEW, text)
    if count == 0:
        raise SystemExit("src/main/trial_point.c: create_aug_jac does not look like v1.0.2 (:63-159)")
    inc = '#include "aug_jac/standard_aug_jac.h"'
    if inc in text:
        text = text.replace(inc, inc + '\n#include "aug_jac/aug_jac_hipfact.h"', 1)
    else:
        text = '#include "aug_jac/aug_jac_hipfact.h"\n' + text
    return text


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("sleqp")
    ap.add_argument("--aug-jac", action="store_true")
    ap.add_argument("--tr", action="store_true")
    ap.add_argument("--mat", action="store_true", help="also copy shim/mat_hipfact.{c,h} (device products of SleqpMat)")
    ap.add_argument("--dry-run", action="store_true")
    args = ap.parse_args()
    sleqp = os.path.abspath(args.sleqp)
    search = os.path.join(sleqp, "cmake", "SearchFact.cmake")
    if not os.path.isfile(search):
        raise SystemExit(f"{search} not found")
    copies, sources = plan(sleqp, args.aug_jac, args.tr, args.mat)
    for src, dst in copies:
        print(("would copy " if args.dry_run else "copy ") + f"{src} -> {dst}")
        if not args.dry_run:
            shutil.copyfile(os.path.join(ROOT, src), os.path.join(sleqp, dst))
    new = register(open(search).read(), sources)
    print(("would register" if args.dry_run else "register") + " add_fact(NAME \"HIPFACT\" ...) in cmake/SearchFact.cmake")
    if not args.dry_run:
        open(search, "w").write(new)
    if args.aug_jac:
        tp = os.path.join(sleqp, "src", "main", "trial_point.c")
        patched = patch_trial_point(open(tp).read())
        print(("would patch" if args.dry_run else "patch") + " create_aug_jac in src/main/trial_point.c")
        if not args.dry_run:
            open(tp, "w").write(patched)
    print(f"now: cmake -DSLEQP_FACT=HIPFACT -DHIPFACT_ROOT={ROOT} <sleqp>")
    return 0


if __name__ == "__main__":
    sys.exit(main())
