#!/usr/bin/env python3
"""Run the REFERENCE's own `test.py` -- the file itself, unmodified, through `runpy` -- on the MI355X path.

    python tools/run_reference_test_py.py --reference-root /path/to/SUPIR [--workdir DIR] [--tokenizer DIR] -- \\
        --img_dir in/ --save_dir out/ --no_llava [--use_tile_vae] [--loading_half_params] [--diff_dtype bf16] ...

Everything after `--` is test.py's own command line (test.py:20-58).  What this launcher does before handing over, and why
(hard part 6):

  1. puts the reference checkout on sys.path and chdir()s to `--workdir` (default: the checkout): test.py opens
     'options/SUPIR_v0.yaml' relative to the working directory (test.py:62).  A workdir with its own `options/SUPIR_v0.yaml`
     (e.g. other checkpoint paths) is how a configuration is changed WITHOUT touching test.py;
  2. `supir_amd.plugin.install()`: every `target:` of that YAML, and `SUPIR.util.PIL2Tensor / Tensor2PIL`, resolve to the HIP-backed
     classes of this package (supir_amd/plugin.py TARGET_MAP) -- test.py, SUPIR/util.py:create_SUPIR_model and the YAML stay as
     they are;
  3. with `--no_llava` on test.py's command line: pre-seeds `llava.llava_agent` with a stub `LLavaAgent` -- test.py imports it
     unconditionally (test.py:5) and the vendored LLaVA does not import under current `transformers` (it re-registers the "llava"
     config name); the stub raises if test.py ever constructs it.  Without `--no_llava` the real module is imported (LLaVA itself
     is out of this package's scope);
  4. `CKPT_PTH` (test.py:6) comes from the checkout; if absent a module with the four path names set to None is pre-seeded;
  5. tokeniser vocabulary: SUPIR_CLIP_TOKENIZER, else `--tokenizer`, else CKPT_PTH.SDXL_CLIP1_PATH when that directory exists (it
     is where the reference itself loads `CLIPTokenizer` from, sgm/modules/encoders/modules.py:462).

Nothing here imports `oracle/` or anything else of the test infrastructure: this is the product's entry point for reference users.
"""
import argparse
import os
import runpy
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _preseed_llava():
    class LLavaAgent:                      # the names test.py:5,72 touch
        def __init__(self, *a, **kw):
            raise RuntimeError("LLaVA is not available in this launcher: run test.py with --no_llava")

    pkg = sys.modules.get("llava") or types.ModuleType("llava")
    pkg.__path__ = getattr(pkg, "__path__", [])
    mod = types.ModuleType("llava.llava_agent")
    mod.LLavaAgent = LLavaAgent
    sys.modules["llava"], sys.modules["llava.llava_agent"] = pkg, mod
    pkg.llava_agent = mod


def _ensure_ckpt_pth(ref_root):
    if "CKPT_PTH" in sys.modules or os.path.exists(os.path.join(ref_root, "CKPT_PTH.py")):
        return
    m = types.ModuleType("CKPT_PTH")
    for k in ("LLAVA_CLIP_PATH", "LLAVA_MODEL_PATH", "SDXL_CLIP1_PATH", "SDXL_CLIP2_CKPT_PTH"):
        setattr(m, k, None)
    sys.modules["CKPT_PTH"] = m


def _tokenizer_dir(explicit):
    if os.environ.get("SUPIR_CLIP_TOKENIZER"):
        return os.environ["SUPIR_CLIP_TOKENIZER"]
    if explicit:
        return explicit
    try:
        import CKPT_PTH
        p = getattr(CKPT_PTH, "SDXL_CLIP1_PATH", None)
        if p and os.path.isdir(p):
            return p
    except Exception:
        pass
    return None


def run(reference_root, test_args, workdir=None, tokenizer=None):
    """Run <reference_root>/test.py as __main__ with `test_args` as its command line; returns runpy's globals dict."""
    reference_root = os.path.abspath(reference_root)
    script = os.path.join(reference_root, "test.py")
    if not os.path.isfile(script):
        raise FileNotFoundError(f"{script}: --reference-root must point at a checkout of Fanghua-Yu/SUPIR")
    for p in (ROOT, reference_root):
        if p not in sys.path:
            sys.path.insert(0, p)
    from supir_amd import plugin
    plugin.install()
    if "--no_llava" in test_args:
        _preseed_llava()
    _ensure_ckpt_pth(reference_root)
    tok = _tokenizer_dir(tokenizer)
    if tok:
        os.environ["SUPIR_CLIP_TOKENIZER"] = tok
    old_argv, old_cwd = sys.argv, os.getcwd()
    sys.argv = [script] + list(test_args)
    os.chdir(workdir or reference_root)
    try:
        return runpy.run_path(script, run_name="__main__")
    finally:
        sys.argv = old_argv
        os.chdir(old_cwd)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    rest = []
    if "--" in argv:
        i = argv.index("--")
        argv, rest = argv[:i], argv[i + 1:]
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--reference-root", default=os.environ.get("SUPIR_REFERENCE_ROOT", os.getcwd()))
    ap.add_argument("--workdir", default=None, help="directory holding options/SUPIR_v0.yaml (default: the reference checkout)")
    ap.add_argument("--tokenizer", default=None, help="directory with the CLIP BPE vocab.json / merges.txt")
    a = ap.parse_args(argv)
    run(a.reference_root, rest, workdir=a.workdir, tokenizer=a.tokenizer)


if __name__ == "__main__":
    main()
