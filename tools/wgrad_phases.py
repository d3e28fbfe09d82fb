"""Phase timing of conv_wgrad1d_h_kernel (library built with -DEMSA_WH_DBG=1, EMSA_LIB pointing at it):
mean shader cycles per K step of each phase, per wave of the workgroup.
usage: EMSA_LIB=tools/bin/whdbg/libemsanet_hip.so python tools/wgrad_phases.py [shape-substring]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from emsanet_amd import _lib, functional as Fn      # noqa: E402
from tools.conv_bench import SHAPES                  # noqa: E402

DEV, DT = 'cuda:0', torch.bfloat16
PH = ('load issue', 'lds read + mfma', 'barrier 1', 'vmcnt + transpose + lds write', 'barrier 2',
      'prologue', 'epilogue', 'total')


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else '1x3 c128'
    n = int(os.environ.get('EMSA_BENCH_N', '32'))
    L = _lib.lib()
    rd = L.emsa_wgrad1d_h_dbg_read
    rd.restype = ctypes.c_int
    rd.argtypes = [ctypes.c_void_p, ctypes.c_int]
    for name, cin, cout, k, s, p, h, w in SHAPES:
        if only not in name:
            continue
        spec = Fn.ConvSpec(cin, cout, k, s, p)
        oh, ow = spec.out_hw(h, w)
        x = Fn.act_empty(n, cin, h, w, DEV, dtype=DT).normal_()
        dy = Fn.act_empty(n, cout, oh, ow, DEV, dtype=DT).normal_()
        wt = torch.randn(cout, cin, *k, device=DEV)
        for _ in range(3):
            Fn.conv_wgrad(x, dy, spec, True, like=wt)
        torch.cuda.synchronize()
        tab = np.zeros(8 * 4 * 4096, dtype=np.int64)
        assert rd(tab.ctypes.data, tab.size) == 0
        tab = tab.reshape(4096, 4, 8)
        nwg = int((tab[:, 0, 7] > 0).sum())
        t = tab[:nwg].astype(np.float64)
        print(f"{name}: {nwg} workgroups, mean total {t[:, :, 7].mean():.0f} cycles per wave")
        for i, ph in enumerate(PH):
            print(f"  {ph:32s}" + ' '.join(f"w{wv} {t[:, wv, i].mean():8.0f}" for wv in range(4)))


if __name__ == '__main__':
    main()
