"""Generate golden fixture G10 (CLI contract) by IMPORTING the reference's arg_utils.py:
defaults of all 55 flags, plus the parse of every rrl_main command line found in the
reference's scripts/navigation1.sh, navigation2.sh and maze.sh.

Run: python tests/golden/gen_cli_golden.py -> tests/golden/cli_golden.json (data only).
"""
import json
import os
import re
import shlex
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_shims  # noqa: E402

_ref_shims.install()
import arg_utils  # noqa: E402  (the reference's)


def parse(argv):
    old = sys.argv
    sys.argv = ["rrl_main"] + argv
    try:
        return vars(arg_utils.get_args())
    finally:
        sys.argv = old


def main():
    out = {"defaults": parse([]), "scripts": []}
    for sh in ("navigation1.sh", "navigation2.sh", "maze.sh"):
        for line in open(os.path.join(_ref_shims.REFERENCE_ROOT, "scripts", sh)):
            m = re.search(r"python -m rrl_main (.*)$", line.strip())
            if not m:
                continue
            argv = shlex.split(m.group(1).replace("$i", "1"))
            out["scripts"].append({"script": sh, "argv": argv, "parsed": parse(argv)})
    json.dump(out, open(os.path.join(HERE, "cli_golden.json"), "w"), indent=1, sort_keys=True)
    print(len(out["defaults"]), "flags;", len(out["scripts"]), "script command lines")


if __name__ == "__main__":
    main()
