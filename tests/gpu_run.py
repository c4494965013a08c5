"""One way for the GPU parity tests to run a configuration: in THIS process through libsimuscop_host.so (`simu_run`, the
body of the `simuReads` command line) -- the HIP context, the engine's device blocks and the table conversions then stay
warm from one case to the next -- or, for a share of the cases, through the command line itself in a child process, so that
`main()`, its option parsing and its exit codes stay covered.  Round 3's suite started a process and a HIP context per
case: 184 of them were 200 s of the 498 s the suite took."""
import contextlib
import os
import subprocess

import simuscop_amd
import simuscop_amd.build as build

SIMU = os.path.join(build.LIBDIR, "simuReads")


@contextlib.contextmanager
def _environ(overrides):
    old = {k: os.environ.get(k) for k in overrides}
    os.environ.update({k: str(v) for k, v in overrides.items()})
    try:
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def run_gpu(cfg, seed, out_dir, host_haplotypes=False, gzip=False, env=None, via_cli=False, timeout=300, flags=()):
    """Returns (ok, message).  `env`: SG_* / SIMU_* knobs of this one run (read by the libraries at run time); `flags`: of
    "crlf_as_lf", "strict_bases", "unique_contigs" (simu_options fields = command-line flags with dashes)."""
    env = env or {}
    if via_cli:
        extra = (["--host-haplotypes"] if host_haplotypes else []) + (["--gzip"] if gzip else []) + ["--" + f.replace("_", "-") for f in flags]
        r = subprocess.run([SIMU, cfg, "--seed", str(seed), "--out", out_dir, "--quiet", *extra], capture_output=True, text=True,
                           timeout=timeout, env=dict(os.environ, **{k: str(v) for k, v in env.items()}))
        return r.returncode == 0, r.stderr[-2000:]
    try:
        with _environ(env):
            simuscop_amd.run_config(cfg, seed=seed, output_dir=out_dir, quiet=1, host_haplotypes=1 if host_haplotypes else 0,
                                    gzip=1 if gzip else 0, **{f: 1 for f in flags})
        return True, ""
    except simuscop_amd.SimuError as e:
        return False, str(e)
