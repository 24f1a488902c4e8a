"""Pack the reference's per-molecule pickles into one shard:
   python tools/pack_qm8.py data/QM8/preprocess/train out/train.lnzq
(file names as written by dataset/get_qm8_data.py:86-96: QM8_preprocess_<split>_*.p)."""
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanczosnet_amd.dataset.packed import convert_reference_pickles  # noqa: E402

if __name__ == '__main__':
  src, dst = sys.argv[1], sys.argv[2]
  files = sorted(glob.glob(os.path.join(src, '*.p')))
  if not files:
    sys.exit('no *.p files under %s' % src)
  n = convert_reference_pickles(files, dst)
  print('%d molecules -> %s (%d bytes)' % (n, dst, os.path.getsize(dst)))
