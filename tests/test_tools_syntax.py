"""The developer tools are part of the evidence trail (profiles/README.md names the command behind every file): they must at least
parse.  Python tools through ast, shell scripts through `bash -n`; nothing is executed and no GPU is needed."""
import ast
import glob
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_python_tools_parse():
    files = sorted(glob.glob(os.path.join(ROOT, "tools", "*.py"))) + [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
    assert len(files) > 10
    for f in files:
        ast.parse(open(f).read(), filename=f)


def test_shell_tools_parse():
    files = sorted(glob.glob(os.path.join(ROOT, "tools", "*.sh")))
    assert files
    for f in files:
        r = subprocess.run(["bash", "-n", f], capture_output=True, text=True)
        assert r.returncode == 0, (f, r.stderr)


def test_profile_script_names_only_tools_that_exist():
    """tools/profile_round.sh is run once per round on a metered box: a renamed or deleted tool must fail here, not there."""
    import re
    txt = open(os.path.join(ROOT, "tools", "profile_round.sh")).read()
    for name in set(re.findall(r"tools/([A-Za-z0-9_]+\.(?:py|sh))", txt)):
        assert os.path.exists(os.path.join(ROOT, "tools", name)), name
