"""Extract the reference's command-line contract (flag names, defaults, types, actions) from the TEXT of its utils.py with
`ast` - nothing is imported or executed - into tests/golden/cli_flags.json.  The fixture is data: the list of flags a job
script may pass, which the build's utils.parse_args must accept with the same defaults."""
import ast
import json
import os
import sys

REF = "/root/reference/utils.py"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cli_flags.json")


def main():
    tree = ast.parse(open(REF).read())
    flags = []
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == "add_argument":
            names = [a.value for a in node.args if isinstance(a, ast.Constant) and isinstance(a.value, str)]
            ent = {"flags": names}
            for kw in node.keywords:
                if kw.arg == "default":
                    ent["default"] = ast.literal_eval(kw.value)
                elif kw.arg == "type" and isinstance(kw.value, ast.Name):
                    ent["type"] = kw.value.id
                elif kw.arg in ("action", "dest", "nargs"):
                    ent[kw.arg] = ast.literal_eval(kw.value)
            flags.append(ent)
    flags.sort(key=lambda e: e["flags"][0])
    json.dump({"source": "utils.py:182-317 (parse_args)", "flags": flags}, open(OUT, "w"), indent=1)
    print(len(flags), "flags ->", OUT)


if __name__ == "__main__":
    sys.dont_write_bytecode = True
    main()
