"""Fixture generator for golden set G10: the op ORDER and literal PARAMETERS of every train / val transform
recipe of the reference, read out of the reference's own class definitions.

TEST INFRASTRUCTURE.  Runs in the build container only (needs /root/reference); writes data, never source text:
``tests/golden/g10_recipes.json`` = {class name: {"bases": [...], "train": [[op, [args], {kwargs}], ...], "val": [...]}}
where args are literals (numbers, tuples) and the few non-literal arguments are reduced to a token:
``"SIZE"`` (self.size), ``"SIZE/0.875"`` (the val Resize target), ``"SIZE0//10"`` (the blur kernel size),
``"BILINEAR"``.  torchvision is not in this image, so the reference module cannot be imported and *called*; its syntax
tree can be read (utils/transforms.py:62-235, utils/util_functions.py:104-109), which pins every number and every ordering
the port's ``RECIPES`` table restates.  The per-op sampling distributions of torchvision 0.5 stay a restatement.

    python oracle/make_golden_recipes.py
"""
import ast
import json
import os

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "g10_recipes.json")


def _token(node):
    """Literal value, or one of the symbolic tokens above."""
    try:
        return ast.literal_eval(node)
    except Exception:
        pass
    src = ast.unparse(node).replace(" ", "")
    table = {
        "self.size": "SIZE",
        "Image.BILINEAR": "BILINEAR",
        "(int(self.size[0]/0.875),int(self.size[1]/0.875))": "SIZE/0.875",
        "self.size[0]//10": "SIZE0//10",
        "constants.IMAGENET_MEAN": "IMAGENET_MEAN",
        "constants.IMAGENET_STD": "IMAGENET_STD",
    }
    if src in table:
        return table[src]
    if isinstance(node, ast.Tuple):
        return [_token(e) for e in node.elts]
    raise ValueError("unreduced argument: %s" % src)


def _op(call):
    name = call.func.attr if isinstance(call.func, ast.Attribute) else call.func.id
    args, kwargs = [], {}
    for a in call.args:
        if isinstance(a, ast.List):                       # RandomApply([op(...)], p=..)
            args.append([_op(e) for e in a.elts])
        else:
            args.append(_token(a))
    for k in call.keywords:
        kwargs[k.arg] = _token(k.value)
    return [name, args, kwargs]


def _compose_ops(fn):
    """The list literal handed to transforms.Compose(...) in the function's return / assignment."""
    for node in ast.walk(fn):
        if isinstance(node, ast.Call) and getattr(node.func, "attr", None) == "Compose":
            return [_op(e) for e in node.args[0].elts]
    return None


def main():
    tree = ast.parse(open(os.path.join(REF, "utils", "transforms.py")).read())
    out = {}
    for cls in tree.body:
        if not isinstance(cls, ast.ClassDef):
            continue
        entry = {"bases": [ast.unparse(b) for b in cls.bases]}
        for fn in cls.body:
            if isinstance(fn, ast.FunctionDef) and fn.name in ("make_train_transform", "make_val_transform"):
                ops = _compose_ops(fn)
                if ops is not None:
                    entry[fn.name.split("_")[1]] = ops
        out[cls.name] = entry
    # RandomGaussianBlur's defaults (utils/util_functions.py:104-109)
    ut = ast.parse(open(os.path.join(REF, "utils", "util_functions.py")).read())
    for cls in ut.body:
        if isinstance(cls, ast.ClassDef) and cls.name == "RandomGaussianBlur":
            init = [f for f in cls.body if isinstance(f, ast.FunctionDef) and f.name == "__init__"][0]
            names = [a.arg for a in init.args.args]
            defaults = [ast.literal_eval(d) for d in init.args.defaults]
            out["RandomGaussianBlur"] = {"init_args": names[1:], "defaults": dict(zip(names[-len(defaults):], defaults))}
    # the normalisation constants the Compose lists name (constants.py)
    ct = ast.parse(open(os.path.join(REF, "constants.py")).read())
    consts = {}
    for node in ct.body:
        if isinstance(node, ast.Assign) and getattr(node.targets[0], "id", "") in ("IMAGENET_MEAN", "IMAGENET_STD"):
            v, mul = node.value, 1.0
            if isinstance(v, ast.BinOp) and isinstance(v.op, ast.Mult):   # np.array([...], dtype=np.float32) * 255
                mul, v = float(ast.literal_eval(v.right)), v.left
            if isinstance(v, ast.Call):
                v = v.args[0]
            consts[node.targets[0].id] = {"values": [float(x) for x in ast.literal_eval(v)], "times": mul}
    out["constants"] = consts
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", os.path.normpath(OUT), {k: list(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
