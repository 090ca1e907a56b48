"""Post-decode re-alignment restated -- test infrastructure.

``alignment2path`` <- wrapper/transcode.py:279-349 (`_alignment2path`): Viterbi search through the soft alignment
matrix, dense, cell by cell, as the reference does it (including the scan that tests every cell against
``min_score`` and numpy's wrap-around of index -1 in the backward pass).
"""
import numpy as np


def alignment2path(alignment, i_max, j_max, min_score):
    fw = np.zeros((i_max, j_max), dtype=np.float32)                      # tc:291
    dist = 0
    i, j = 0, 0
    while i < i_max and j < j_max:                                       # tc:294
        im1 = fw[i - 1, j] if i > 0 else 0
        jm1 = fw[i, j - 1] if j > 0 else 0
        ijm1 = fw[i - 1, j - 1] if i > 0 and j > 0 else 0
        fw[i, j] = alignment[j][i] + max(im1, jm1, ijm1)                 # tc:307
        while True:                                                      # tc:308-316: next cell above min_score
            i += 1
            if i == i_max:
                j += 1
                if j == j_max:
                    break
                i = 0
            if alignment[j][i] > min_score:
                break
    i = i_max - 1 if i_max <= j_max else j_max - 2 + int(np.argmax(fw[j_max - i_max - 2:, j_max - 1]))   # tc:318-321
    j = j_max - 1 if j_max <= i_max else i_max - 2 + int(np.argmax(fw[i_max - 1, i_max - j_max - 2:]))
    realignment = {i_max: j_max}
    while i >= 0 and j >= 0:                                             # tc:323-337
        dist += 1.0 - alignment[j][i]
        realignment[i] = j
        if fw[i - 1, j] > fw[i, j - 1]:
            if fw[i - 1, j] > fw[i - 1, j - 1]:
                i -= 1
            else:
                i -= 1
                j -= 1
        elif fw[i, j - 1] > fw[i - 1, j - 1]:
            j -= 1
        else:
            j -= 1
            i -= 1
    realignment[0] = 0
    return realignment, dist
