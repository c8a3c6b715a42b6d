"""``python -m omg_amd.run <script.py> [script arguments ...]`` — run one of the reference's scripts UNCHANGED on this backend (B3).

    python -m omg_amd.run /path/to/OMG/inference_lora.py --prompt "..." --lora_path "..." ...
    python -m omg_amd.run /path/to/OMG/inference_instantid.py ...

What it does, in this order: the script's own directory in front of ``sys.path`` (as ``python script.py`` would put it, so that the
checkout's other ``src.*`` packages — detectors, segmenters — resolve to the reference's own files WHATEVER the current directory is: the
alias package ``src`` that ``install()`` registers lists the ``src`` directories found on ``sys.path`` at that moment as its ``__path__``),
``omg_amd.compat.install()`` (alias modules for ``src.pipelines.*``, ``src.prompt_attention.p2p_attention`` and ``diffusers``:
INTEGRATION.md §1), ``sys.argv`` = the script and its arguments, then ``runpy.run_path(script, run_name="__main__")``.  Nothing of the
script is edited or copied."""
import os
import runpy
import sys


def main(argv=None) -> None:
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] in ("-h", "--help"):
        print(__doc__)
        raise SystemExit(0 if argv else 2)
    script = os.path.abspath(argv[0])
    if not os.path.isfile(script):
        raise SystemExit(f"omg_amd.run: no such script: {argv[0]}")
    from . import compat
    sys.path.insert(0, os.path.dirname(script))          # BEFORE install(): its stand-in for `src` must see the checkout's own src/ directory
    registered = compat.install()
    sys.argv = [script] + argv[1:]
    try:
        runpy.run_path(script, run_name="__main__")
    finally:
        compat.uninstall()
        del registered


if __name__ == "__main__":
    main()
