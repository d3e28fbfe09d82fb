"""Static checks of the compiled gfx950 kernels (cross-compiled, no GPU needed): no scratch
(register spills) in any kernel, and no `s_waitcnt` that blocks an MFMA kernel's K loop on
freshly issued global loads before the step's matrix work -- both were real, silent 5-40 %
regressions during development."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
needs_hipcc = pytest.mark.skipif(not (shutil.which('hipcc') or os.path.exists('/opt/rocm/bin/hipcc')),
                                 reason='hipcc not available')


@needs_hipcc
def test_no_early_waits_in_mfma_loops():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'check_early_waits.py')],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert 'early waits: 0' in r.stdout


@needs_hipcc
def test_no_register_spills():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'check_spills.py')],
                       capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@needs_hipcc
def test_conv_rs_has_no_vmcnt_wait_inside_its_mfma_sequence():
    """csrc/conv_rs.hip: a compiler-placed `s_waitcnt vmcnt` between the first and the last MFMA of a
    tile drains the next tile's loads in every iteration (found in round 4: the waits for the
    resident weight fragments sat there until the fragments were pinned in the prologue)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'check_rs_waits.py')],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and 'conv_rs wait check: ok' in r.stdout, r.stdout + r.stderr


@needs_hipcc
def test_wgrad16_kernels_use_the_lds_instructions_their_layouts_are_built_for():
    """csrc/conv_mfma.hip: conv_wgrad1d_h_kernel's 136-byte rows are conflict-free for ds_read_b64 /
    ds_write_b64 only -- fused into ds_read2_b64 / ds_write2_b64 (load-store optimizer, or the IR
    vectorizer: it happened silently in round 4) they fall back to the 32-bank rule at four times
    the cycles; conv_wgrad1d_tr_kernel must read its fragments with ds_read_b64_tr_b16 and carry no
    register transposes or shifts"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'check_wgrad16_isa.py')],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and 'wgrad16 isa check: ok' in r.stdout, r.stdout + r.stderr


@needs_hipcc
def test_streaming_kernels_issue_their_loads_together():
    """csrc/pointwise.hip (round 5): the BatchNorm passes, the up-sampling / max-pool
    forward, the SE scale kernels and channel_dot must not wait for a load before the next one is
    issued -- an optional operand behind a run-time branch, or a border test around a load, puts
    `s_waitcnt vmcnt(0)` behind every load (one memory round trip each: the round-4 forms had 3-13
    per element).  tools/isa_audit.py counts those load -> wait pairs per kernel."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'isa_audit.py'),
                        os.path.join(ROOT, 'emsanet_amd', 'csrc', 'pointwise.hip'), '--min-ser', '0'],
                       capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    watched = ('bn_act_fwd_fast_kernel', 'bn_bwd_reduce_fast_kernel', 'bn_bwd_apply_fast_kernel',
               'up2x_dw_fwd_kernel', 'maxpool_fwd_kernel', 'se_scale_add_fwd_kernel',
               'se_scale_bwd_apply_kernel', 'channel_dot_kernel')
    seen, bad = set(), []
    for line in r.stdout.splitlines()[1:]:
        f = line.split()
        if len(f) < 5:
            continue
        kname = ' '.join(f[4:])
        for w in watched:
            if w in kname:
                seen.add(w)
                # (<= 3: the prologue of bn_bwd_apply -- the slice-sum merge -- and single tail loads)
                if int(f[0]) > 3:
                    bad.append(line)
    assert seen == set(watched), set(watched) - seen
    assert not bad, '\n'.join(bad)
