import sys
import numpy as np
a = np.load(sys.argv[1]); b = np.load(sys.argv[2])
bad = np.flatnonzero(np.any(a["U"].reshape(len(a["old"]), -1) != b["U"].reshape(len(b["old"]), -1), axis=1))
print("columns differing in U:", len(bad), bad[:40])
print(" old a/b:", a["old"][bad][:20], b["old"][bad][:20])
print(" npass a/b:", a["npass"][bad][:20], b["npass"][bad][:20])
