"""PER scoring chain downstream of the decoder -- host-side, pure Python.

Restates, with the same results on the same inputs (golden vectors in tests/golden/scoring.json):
  ``loadMlf``          reference scripts/util/kw_utils.py:45-77
  ``convert_results``  reference scripts/util/evaluateModelForTest.py:13-34   (60 -> 39 phone map on the recogniser MLF)
  ``levenshtein_nist`` reference scripts/util/levenshtein.py:84-164         (TIMIT/NIST alignment: 'sil' is optional)
  ``computeWER``       reference scripts/util/levenshtein.py:166-205
"""
from __future__ import annotations


def loadMlf(filename):
    """HTK master label file -> {utterance id: [words, begins, ends, order index]}.

    Quoted lines open an utterance (id = text up to the first '.', a leading ``*/`` dropped); a line starting with '.'
    closes it; lines after the closing dot are skipped until the next quoted line; three-token lines are
    ``begin end word``; any other line is a bare word (a blank line yields an empty word, as in the reference)."""
    mlf = {}
    order = 0
    utt = None
    words, begins, ends = [], [], []
    closed = False
    with open(filename) as fid:
        for line in fid.readlines():
            if not line or line[0] == "#":
                continue
            if line[0] == '"':
                utt = line[1:].split(".")[0]
                if utt[0] == "*":
                    utt = utt[2:]
                words, begins, ends = [], [], []
                closed = False
                continue
            if line[0] == ".":
                mlf[utt] = [words, begins, ends, order]
                order += 1
                closed = True
                continue
            if closed:
                continue
            tok = line.split()
            if len(tok) == 3:
                begins.append(int(tok[0]))
                ends.append(int(tok[1]))
                words.append(tok[2])
            else:
                words.append(line.strip())
    return mlf


def convert_results(file_phone_map, file_in, file_out):
    """Map the recogniser's 60-phone output to the 39-phone scoring set; ``.lab`` -> ``.rec`` in utterance headers."""
    phone_map = {}
    with open(file_phone_map) as fid:
        for entry in fid.readlines():
            cols = entry.strip().split()
            if len(cols) == 3:
                phone_map[cols[1]] = cols[2]
    with open(file_in) as fin, open(file_out, "w") as fout:
        for raw in fin.readlines():
            text = raw.strip()
            if text and text[0] == '"':
                text = text.replace(".lab", ".rec").replace("\\", "/")
            cols = text.split()
            if cols and cols[0][0] not in ("#", ".", '"'):
                cols[-1] = phone_map[cols[-1]]
                text = " ".join(cols)
            fout.write(text + "\n")


_SUB, _INS, _DEL = 0, 1, 2


def levenshtein_nist(hyp, ref, _flip=False):
    """-> (N, H, D, I, S) with the reference's conventions: the longer sequence drives the outer loop (``_flip`` marks a
    swap), 'sil' in the reference may be skipped for free, ties prefer the move that can absorb an optional 'sil', and
    the back-trace stops when either sequence is exhausted (a leading remainder is not counted)."""
    s1, s2 = hyp, ref
    if len(s1) < len(s2):
        return levenshtein_nist(s2, s1, True)
    if len(s2) == 0:
        # the reference returns a bare int here and its caller then fails to unpack; score it as all-insertions instead
        n = len(s1)
        return (n, 0, n, 0, 0) if _flip else (0, 0, 0, n, 0)
    n1, n2 = len(s1), len(s2)
    move = [[0] * n2 for _ in range(n1)]
    prev = list(range(n2 + 1))
    for i in range(n1):
        cur = [i + 1]
        for j in range(n2):
            ins = prev[j + 1] + 1
            dele = cur[j] + 1
            sub = prev[j] + (s1[i] != s2[j])
            if _flip and s1[i] == "sil":
                ins -= 1
            if (not _flip) and s2[j] == "sil":
                dele -= 1
            best = min(ins, dele, sub)
            cur.append(best)
            m = _SUB if sub == best else (_INS if ins == best else _DEL)  # argmin over [sub, ins, del]
            if m != _SUB and ins == dele:
                m = _INS if _flip else _DEL
            move[i][j] = m
        prev = cur
    i, j = n1 - 1, n2 - 1
    H = D = I = S = skips = 0
    while i >= 0 and j >= 0:
        m = move[i][j]
        if m == _SUB:
            if s1[i] == s2[j]:
                H += 1
            else:
                S += 1
            i -= 1
            j -= 1
        elif m == _INS:
            if _flip and s1[i] == "sil":
                skips += 1
            else:
                I += 1
            i -= 1
        else:
            if (not _flip) and s2[j] == "sil":
                skips += 1
            else:
                D += 1
            j -= 1
    if _flip:
        I, D = D, I
        N = n1
    else:
        N = n2
    return N, H + skips, D, I, S


def levenshtein2(hyp, ref, _flip=False):
    """Plain alignment counts (reference levenshtein.py:26-81): argmin over [ins, del, sub], no optional silence."""
    s1, s2 = hyp, ref
    if len(s1) < len(s2):
        return levenshtein2(s2, s1, True)
    if len(s2) == 0:
        n = len(s1)
        return (n, 0, n, 0, 0) if _flip else (0, 0, 0, n, 0)
    n1, n2 = len(s1), len(s2)
    move = [[0] * n2 for _ in range(n1)]
    prev = list(range(n2 + 1))
    for i in range(n1):
        cur = [i + 1]
        for j in range(n2):
            ins = prev[j + 1] + 1
            dele = cur[j] + 1
            sub = prev[j] + (s1[i] != s2[j])
            best = min(ins, dele, sub)
            cur.append(best)
            move[i][j] = 0 if ins == best else (1 if dele == best else 2)
        prev = cur
    i, j = n1 - 1, n2 - 1
    H = D = I = S = 0
    while i >= 0 and j >= 0:
        m = move[i][j]
        if m == 0:
            I += 1
            i -= 1
        elif m == 1:
            D += 1
            j -= 1
        else:
            if s1[i] == s2[j]:
                H += 1
            else:
                S += 1
            i -= 1
            j -= 1
    if _flip:
        I, D = D, I
        N = n1
    else:
        N = n2
    return N, H, D, I, S


def computeWER(testMlfFile, refMlfFile, NISTvariant=False):
    """100 * (D + I + S) / N over all test utterances, visited in their original file order."""
    test = loadMlf(testMlfFile) if isinstance(testMlfFile, str) else testMlfFile
    ref = loadMlf(refMlfFile) if isinstance(refMlfFile, str) else refMlfFile
    TD = TS = TI = TN = 0
    for utt in sorted(test.keys(), key=lambda u: (test[u][3], u)):
        if utt not in ref:
            raise KeyError("Utterance " + utt + " not in the reference mlf")
        align = levenshtein_nist if NISTvariant else levenshtein2
        N, _H, D, I, S = align(test[utt][0], ref[utt][0])
        TD += D
        TS += S
        TI += I
        TN += N
    TN += (TN == 0)
    return (100.0 * (TD + TI + TS)) / TN
