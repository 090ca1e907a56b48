"""Error-rate metrics of `Sequence2Sequence.evaluate()` (seq2seq.py:651-754), without the reference package and
without its third-party dependencies (`uniseg`, `rapidfuzz` are not needed).

What the reference computes (`lib/alignment.py:140-486`), restated here:

* `Alignment` -- difflib (Ratcliff-Obershelp) opcodes between two strings or two word lists turned into a list of
  (source, target) pairs with a gap element on the shorter side of every non-equal block (`alignment.py:165-204`);
  for strings, combining marks are then re-attached to the preceding base letter on either side so that a grapheme
  cluster counts as one symbol (`:206-240`); optionally a confusion table of non-identical pairs, where pairs next to
  gaps are merged into multi-character entries (`:242-259,261-278`).
* `get_adjusted_distance` -- normalisation of both sides (NFC / NFKC / "historic_latin" ligature and PUA
  replacements when `gtlevel < 3`), alignment, and the count of pairs that are neither identical nor equivalent
  (for "historic_latin" at `gtlevel == 1`: long s / round s, umlaut spellings, quote and dash variants ...)
  (`:290-357`).
* `Edits` -- running length-weighted mean and variance of per-line rates (Chan et al. 1979 pairwise update),
  token histograms of both sides and the worst lines (`:367-459`).
* `splitwords` -- Unicode word segmentation (UAX #29) without the pure punctuation / space / symbol tokens
  (`:461-486`).  The reference delegates the segmentation to `uniseg`; CPython ships no Word_Break property table, so
  the classes are derived here from the General_Category plus the explicit MidLetter / MidNum / MidNumLet / quote
  code points of UAX #29.  For alphabetic scripts with European punctuation (the domain of this tool) that yields the
  same tokens; scripts that need dictionaries or per-script tables (Thai, CJK ideographs) come out one character per
  token.

Deliberate divergences from the reference's EFFECTIVE behaviour (both pinned in tests/test_metrics.py):
  * 'historic_latin' normalisation.  The reference's `normalize()` pops the single-code-point keys out of the module-global
    `L2_HISTLAT_EQV` while it builds its translation table (alignment.py:318-320 `equivalences.pop(key)`), so only the very
    FIRST string normalised in a process gets the single-character replacements (ligatures such as U+FB01, PUA letters,
    macron -> tilde); every later call applies the multi-code-point replacements only.  `normalize_text` applies the whole
    table on every call -- what the table is evidently for.  On text containing such characters the CER/WER of `evaluate()`
    therefore differ from figures produced by the reference; `reference_quirks(True)` (or CASV_METRICS_REFERENCE_QUIRKS=1 in
    the environment) reproduces the reference's process-global pop so that published numbers stay comparable.
  * `splitwords` derives Word_Break classes from General_Category instead of uniseg's tables (above).
"""
import logging
import os
import threading
import unicodedata
from bisect import bisect_left, insort_left
from difflib import SequenceMatcher
from itertools import chain

# ---------------------------------------------------------------------------------------------------------------------
# "historic_latin" tables (alignment.py:8-137).  Level 1: classes of single symbols that count as equal.
# ---------------------------------------------------------------------------------------------------------------------
# Written with escapes throughout: several members are private-use or combining code points that editors mangle.
_L1_CLASSES = [
    ['\u00e4', 'a\u0308', 'a\u0364'],
    ['\u00f6', 'o\u0308', 'o\u0364'],
    ['\u00fc', 'u\u0308', 'u\u0364'],
    ['\u00c4', 'A\u0308', 'A\u0364'],
    ['\u00d6', 'O\u0308', 'O\u0364'],
    ['\u00dc', 'U\u0308', 'U\u0364'],
    's\u017f',
    'r\ua75b',
    'z\u0292',
    'Z\u01b7',
    'n\u019e',
    '\u03bc\u00b5',
    '\u03c0\U0001d6d1\U0001d70b\U0001d745\U0001d77f\U0001d7b9',
    '0\u2070',
    '1\u00b9',
    '2\u00b2',
    '3\u00b3',
    '4\u2074',
    '5\u2075',
    '6\u2076',
    '7\u2077',
    '8\u2078',
    '9\u2079\ua770',
    '\u201e\u00bb\u203a\u301f',
    '\u201c\u00ab\u2039\u301e',
    '\'\u02b9\u02bc\u2032\u2018\u2019\u201b\u1fbd`',
    ',\u201a',
    '-\u2212\u2014\u2010\u2011\u2012\u2013\u2043\ufe58\u2015\u2500\u2e17',
    '\u201f\u3003\u201d\u2033',
    '~\u223c\u02dc\u1fc0\u2053',
    '(\u27e8\u207d',
    ')\u27e9\u207e',
    '/\u29f8\u2044\u2215',
    '\\\u29f9\u2216\u29f5',
]
L1_EQUIVALENCES = [set(members) for members in _L1_CLASSES]

# Level 2: ligatures, private-use code points and abbreviation marks replaced before aligning (key -> replacement).
# Where the reference's literal lists a key twice (U+EEDC, U+E8BF, U+E8B7), the later entry is the effective one.
_L2_PAIRS = [
    ('\uf502', 'ch'), ('\ueec4', 'ck'), ('\ufb05', '\u017ft'), ('\ufb01', 'fi'), ('\ufb00', 'ff'),
    ('\ufb02', 'fl'), ('\ufb03', 'ffi'), ('\uf4fc', '\u017fk'), ('\ueedc', 't\u0292'), ('\uf532', 'as'),
    ('\uf533', 'is'), ('\uf534', 'us'), ('\uf535', 'Qu'), ('\u0133', 'ij'), ('\ue8bf', 'q\u0292'),
    ('\ueba5', '\u017fp'), ('\ufb06', 'st'), ('q\u0308', 'q\u1dd3'), ('c\u0308', 'c\u1dd3'),
    ('\u1e21', 'g\u1dd3'), ('v\u0309', 'v\u1de3'), ('v\u1dce', 'v\u1de3'), ('b\u1dce', 'b\u1de3'),
    ('p\u1dce', 'p\u1de3'), ('d\u0309', '\u00f0'), ('\ua75f', 'v\u1de3'), ('t\u1de3', 't\u1dd1'),
    ('\ueada', '\u017ft'), ('\ueba2', '\u017fi'), ('\ueba3', '\u017fl'), ('\ueba6', '\u017f\u017f'),
    ('\ueba7', '\u017f\u017fi'), ('\uf4ff', '\u017f\u017ft'), ('\uf52c', '\u017fp'), ('\ueec5', 'ct'),
    ('\ueecb', 'ft'), ('\ue5d2', 'm\u0303'), ('\ue5dc', '\u00f1'), ('\ue665', 'p\u0303'), ('\ue42c', 'a\u0364'),
    ('\ue644', 'o\u0364'), ('\ue72b', 'u\u0364'), ('\ue72d', '\u016f'), ('\uebac', '\u00df'),
    ('\ue8b7', '\u017f\u1de3'), ('\uf1a6', '\ua770'), ('\uf223', 'm'), ('\uf158', '\u204a'),
    ('\uf159', '\u00f0'), ('\uf160', ':'), ('q\uf02f', 'q\u0365'), ('t\uf1cc', 't\u1dd1'), ('\uf4f9', 'll'),
    ('\u0101', 'a\u0303'), ('\u0113', '\u1ebd'), ('\u012b', '\u0129'), ('\u014d', '\u00f5'),
    ('\u016b', '\u0169'), ('c\u0304', 'c\u0303'), ('q\u0304', 'q\u0303'), ('r\u0304', 'r\u0303'),
    ('\uf50e', 'q\u0301'),
]
L2_REPLACEMENTS = dict(_L2_PAIRS)


# Compatibility with the reference's effective behaviour (module docstring): [enabled, single-character keys already popped]
_QUIRKS = [os.environ.get('CASV_METRICS_REFERENCE_QUIRKS', '') == '1', False]
_QUIRKS_LOCK = threading.Lock()          # (normalize_text is reachable from worker threads)


def reference_quirks(enable=True):
    """Reproduce (True) or not (False, default) the reference's process-global pop of the single-code-point replacements
    (alignment.py:318-320); switching it on starts a fresh "process": the next string normalised is the first one."""
    with _QUIRKS_LOCK:
        _QUIRKS[0], _QUIRKS[1] = bool(enable), False


def normalization_mode():
    """Which of the two behaviours `normalize_text('historic_latin')` has right now -- evaluate() says so beside its figures."""
    return 'reference-quirks' if _QUIRKS[0] else 'table-applied-to-every-string'


def normalize_text(seq, normalization=None, gtlevel=1):
    """alignment.py:309-326.  Lists (word tokens) are normalised element-wise."""
    if isinstance(seq, list):
        return [normalize_text(s, normalization, gtlevel) for s in seq]
    if normalization in ('NFC', 'NFKC'):
        return unicodedata.normalize(normalization, seq)
    if normalization == 'historic_latin':
        if gtlevel >= 3:
            return seq
        single = {k: v for k, v in L2_REPLACEMENTS.items() if len(k) == 1}
        if _QUIRKS[0]:
            with _QUIRKS_LOCK:           # test-and-set: exactly one string is "the first one"
                first, _QUIRKS[1] = not _QUIRKS[1], True
            if not first:
                single = {}              # the reference popped them from its global table during the first call
        for key, value in L2_REPLACEMENTS.items():
            if len(key) > 1:
                seq = seq.replace(key, value)        # multi-code-point keys first, in table order
        return seq.translate(str.maketrans(single))
    return seq


def make_equivalence(normalization=None, gtlevel=1):
    classes = L1_EQUIVALENCES if (normalization == 'historic_latin' and gtlevel == 1) else []

    def equivalent(x, y):
        if isinstance(x, list):
            return len(x) == len(y) and all(equivalent(a, b) for a, b in zip(x, y))
        if x == y:
            return True
        return any(x in cls and y in cls for cls in classes)
    return equivalent


# ---------------------------------------------------------------------------------------------------------------------
def _is_letter_start(sym):
    return unicodedata.category(sym[0])[0] == 'L'


class Alignment(object):
    """Pairwise alignment of two strings or two token lists (API of alignment.py:140-365)."""

    def __init__(self, gap_element=0, logger=None, confusion=False):
        self.confusion = {} if confusion else None
        self.gap_element = gap_element
        self.logger = logger or logging.getLogger(__name__)
        self.matcher = SequenceMatcher(isjunk=None, autojunk=False)
        self.source_text = []
        self.target_text = []

    def set_seqs(self, source_text, target_text):
        self.matcher.set_seqs(source_text, target_text)
        self.source_text, self.target_text = source_text, target_text

    def is_bad(self):
        """Hopeless pair: difflib's quick similarity bound below 0.5 on more than 5 symbols (alignment.py:160-163)."""
        return bool(self.matcher.quick_ratio() < 0.5 and len(self.source_text) > 5)

    def _pairs_from_opcodes(self):
        gap, src, tgt = self.gap_element, self.source_text, self.target_text
        pairs = []
        i1 = j1 = 0
        for op, i0, i1, j0, j1 in self.matcher.get_opcodes():
            a, b = list(src[i0:i1]), list(tgt[j0:j1])
            if op == 'equal':
                pairs.extend(zip(a, b))
            elif op in ('replace', 'insert', 'delete'):
                # pair the common length, the surplus of the longer side stands against gaps
                n = min(len(a), len(b))
                pairs.extend(zip(a[:n], b[:n]))
                pairs.extend((s, gap) for s in a[n:])
                pairs.extend((gap, t) for t in b[n:])
            else:
                raise Exception('difflib returned invalid opcode', op, 'in', src, tgt)
        assert i1 == len(src), 'alignment does not span full sequence "%s" - %d' % (src, i1)
        assert j1 == len(tgt), 'alignment does not span full sequence "%s" - %d' % (tgt, j1)
        return pairs

    def _recombine(self, pairs):
        """Combining marks join the symbol before them on the same side when that is a letter; the other side of the
        mark's own pair stays behind against a gap, or joins likewise (alignment.py:206-240)."""
        gap = self.gap_element
        out, changed = [], False

        def joins(sym, side):
            return (sym != gap and unicodedata.combining(sym) and out and out[-1][side] != gap
                    and _is_letter_start(out[-1][side]))
        for s, t in pairs:
            if joins(s, 0):
                out[-1][0] += s
                changed = True
                if t == gap:
                    continue
                if unicodedata.combining(t) and out[-1][1] != gap and _is_letter_start(out[-1][1]):
                    out[-1][1] += t
                    continue
                s = gap
            elif joins(t, 1):
                out[-1][1] += t
                changed = True
                if s == gap:
                    continue
                t = gap
            out.append([s, t])
        return [tuple(p) for p in out] if changed else pairs

    def _count_confusion(self, pairs, eq):
        gap = self.gap_element
        text = lambda x: '' if x == gap else x
        for pos, pair in enumerate(pairs):
            if gap in pair:
                continue
            while pos and gap in pairs[pos - 1]:           # merge the gapped neighbours before it into this entry
                pos -= 1
                pair = (text(pairs[pos][0]) + text(pair[0]), text(pairs[pos][1]) + text(pair[1]))
            if eq and eq(*pair):
                continue
            self.confusion[pair] = self.confusion.get(pair, 0) + 1

    def get_best_alignment(self, eq=None):
        try:
            pairs = self._pairs_from_opcodes()
        except AssertionError:
            raise
        except Exception:
            self.logger.exception('alignment of "%s" and "%s" failed', self.source_text, self.target_text)
            raise
        if not isinstance(self.source_text, list):
            pairs = self._recombine(pairs)
        if self.confusion is not None:
            self._count_confusion(pairs, eq)
        return pairs

    def get_confusion(self, limit=None):
        """([(count, (source, target)), ...] most frequent first, total number of counted pairs)."""
        if self.confusion is None:
            raise Exception('aligner was not configured to count confusion')
        keys, table, total = [], [], 0
        for pair, count in self.confusion.items():
            total += count
            if pair[0] == pair[1]:
                continue
            n = len(table)
            idx = bisect_left(keys, -count, hi=min(limit or n, n))
            if limit and idx >= limit:
                continue
            keys.insert(idx, -count)
            table.insert(idx, (count, pair))
        return (table[:limit] if limit else table), total

    @staticmethod
    def get_levenshtein_distance(source_text, target_text):
        """Plain unit-cost edit distance between code points and the longer length (alignment.py:280-288)."""
        prev = list(range(len(target_text) + 1))
        for i, s in enumerate(source_text, 1):
            cur = [i]
            for j, t in enumerate(target_text, 1):
                cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (s != t)))
            prev = cur
        return prev[-1], max(len(target_text), len(source_text))

    def get_adjusted_distance(self, source_text, target_text, normalization=None, gtlevel=1, return_alignment=False):
        """(number of non-equivalent pairs, alignment length[, alignment]) after normalisation (alignment.py:290-357)."""
        equivalent = make_equivalence(normalization, gtlevel)
        self.set_seqs(normalize_text(source_text, normalization, gtlevel), normalize_text(target_text, normalization, gtlevel))
        alignment = self.get_best_alignment(eq=equivalent)
        dist = float(sum(1 for s, t in alignment if not (s == t or equivalent(s, t))))
        if return_alignment:
            return dist, len(alignment), alignment
        return dist, len(alignment)

    @staticmethod
    def best_alignment(source_text, target_text, with_confusion=False):
        aligner = Alignment(confusion=with_confusion)
        aligner.set_seqs(source_text, target_text)
        if with_confusion:
            return aligner.get_best_alignment(), aligner.get_confusion()
        return aligner.get_best_alignment()


# ---------------------------------------------------------------------------------------------------------------------
class Edits(object):
    """Aggregate of per-line error rates: length-weighted mean, variance, histograms, worst lines (alignment.py:367-459)."""

    class Example(object):
        def __init__(self, mean=0, length=0, name=''):
            self.mean, self.length, self.name = mean, length, name

        def cost(self):
            return -self.mean * self.length          # worst first

        def __repr__(self):
            return ('%s: ' % self.name if self.name else '') + 'avg=%s len=%s' % (self.mean, self.length)

        def __lt__(self, other): return self.cost() < other.cost()
        def __le__(self, other): return self.cost() <= other.cost()
        def __eq__(self, other): return self.cost() == other.cost()
        def __ne__(self, other): return self.cost() != other.cost()
        def __gt__(self, other): return self.cost() > other.cost()
        def __ge__(self, other): return self.cost() >= other.cost()
        __hash__ = object.__hash__

    def __init__(self, logger=None, histogram=False):
        self.logger = logger or logging.getLogger(__name__)
        self.length = 0
        self.mean = 0
        self.varia = 0
        self.score = 0
        self.steps = 0
        # a histogram that is switched on starts with the empty token (and is therefore truthy, see add())
        self.hist1 = {'': 0} if histogram else {}
        self.hist2 = {'': 0} if histogram else {}
        self.worst = []

    def __repr__(self):
        return 'N=%d µ=%.2f σ²=%.2f' % (self.length, self.mean, self.varia)

    def hist(self):
        return {key: (self.hist1.get(key, 0), self.hist2.get(key, 0)) for key in sorted(set(self.hist1) | set(self.hist2))}

    def update(self, steps, length, mean, varia, hist1, hist2):
        """Pairwise combination of (count, mean, variance) aggregates (Chan, Golub, LeVeque 1979), weights = lengths."""
        if length < 1:
            return
        self.steps += steps
        n, m = length, self.length
        delta = mean - self.mean
        self.mean = (n * mean + m * self.mean) / (n + m)
        self.varia = (n * varia + m * self.varia + delta ** 2 * n * m / (n + m)) / (n + m)
        self.length = n + m
        for tok, cnt in hist1.items():
            self.hist1[tok] = self.hist1.get(tok, 0) + cnt
        for tok, cnt in hist2.items():
            self.hist2[tok] = self.hist2.get(tok, 0) + cnt

    def _cap(self):
        return max(int(self.steps * 0.01), 10)          # the worst 1 % of the lines, at least 10

    def add(self, dist, length, seq1, seq2, name=None):
        hist1, hist2 = {}, {}
        if self.hist1:
            for tok in seq1:
                hist1[tok] = hist1.get(tok, 0) + 1
        if self.hist2:
            for tok in seq2:
                hist2[tok] = hist2.get(tok, 0) + 1
        rate = dist / length if length else 0
        self.update(1, length, rate, 0, hist1, hist2)
        insort_left(self.worst, Edits.Example(mean=rate, length=length, name=name))
        self.worst = self.worst[:self._cap()]

    def merge(self, edits, name_prefix=None):
        self.update(edits.steps, edits.length, edits.mean, edits.varia, edits.hist1, edits.hist2)
        if name_prefix:
            for example in edits.worst:
                example.name = name_prefix + (example.name or '')
        self.worst = sorted(chain(self.worst, edits.worst))[:self._cap()]


# ---------------------------------------------------------------------------------------------------------------------
# Word segmentation: default rules of UAX #29 over Word_Break classes derived from General_Category
# ---------------------------------------------------------------------------------------------------------------------
_MIDNUMLET = set('.\u2018\u2019\u2024\ufe52\uff07\uff0e')
_MIDLETTER = set(':\u00b7\u0387\u055f\u05f4\u2027\ufe13\ufe55\uff1a')
_MIDNUM = set(',;\u037e\u0589\u060c\u060d\u066c\u07f8\u2044\ufe10\ufe14\ufe50\ufe54\uff0c\uff1b')
_NEWLINE = set('\u000b\u000c\u0085\u2028\u2029')
_KATAKANA_EXTRA = set('\u3031\u3032\u3033\u3034\u3035\u309b\u309c\u30a0\u30fc\uff70')
_NO_ALETTER_PREFIXES = ('CJK ', 'HIRAGANA', 'THAI ', 'LAO ', 'MYANMAR', 'KHMER')
_class_cache = {}


def _word_break_class(c):
    cls = _class_cache.get(c)
    if cls is not None:
        return cls
    cat = unicodedata.category(c)
    if c == '\r':
        cls = 'CR'
    elif c == '\n':
        cls = 'LF'
    elif c in _NEWLINE:
        cls = 'Newline'
    elif c == '\u200d':
        cls = 'ZWJ'
    elif c == "'":
        cls = 'Single_Quote'
    elif c == '"':
        cls = 'Double_Quote'
    elif c in _MIDNUMLET:
        cls = 'MidNumLet'
    elif c in _MIDLETTER:
        cls = 'MidLetter'
    elif c in _MIDNUM:
        cls = 'MidNum'
    elif cat in ('Mn', 'Me', 'Mc'):
        cls = 'Extend'
    elif cat == 'Cf' and c != '\u200b':
        cls = 'Format'
    elif cat == 'Pc':
        cls = 'ExtendNumLet'
    elif cat == 'Nd':
        cls = 'Numeric'
    elif '\U0001f1e6' <= c <= '\U0001f1ff':
        cls = 'Regional_Indicator'
    elif cat[0] == 'L' or cat == 'Nl':
        name = unicodedata.name(c, '')
        if name.startswith('KATAKANA') or c in _KATAKANA_EXTRA:
            cls = 'Katakana'
        elif name.startswith('HEBREW LETTER') or name.startswith('HEBREW LIGATURE'):
            cls = 'Hebrew_Letter'
        elif cat == 'Lo' and name.startswith(_NO_ALETTER_PREFIXES):
            cls = 'Other'
        else:
            cls = 'ALetter'
    elif cat == 'Zs':
        cls = 'WSegSpace'
    else:
        cls = 'Other'
    _class_cache[c] = cls
    return cls


_AH = ('ALetter', 'Hebrew_Letter')
_MIDQ_LETTER = ('MidLetter', 'MidNumLet', 'Single_Quote')
_MIDQ_NUM = ('MidNum', 'MidNumLet', 'Single_Quote')
_SKIP = ('Extend', 'Format', 'ZWJ')


def word_boundaries(text):
    """Indices i (0 < i < len(text)) at which UAX #29 breaks between text[i-1] and text[i]."""
    n = len(text)
    cls = [_word_break_class(c) for c in text]
    # WB4: Extend / Format / ZWJ are transparent (take part in no rule) unless they follow a line break or start the text
    breaks = []
    for i in range(1, n):
        a, b = cls[i - 1], cls[i]
        if a == 'CR' and b == 'LF':
            continue                                        # WB3
        if a in ('CR', 'LF', 'Newline') or b in ('CR', 'LF', 'Newline'):
            breaks.append(i)                                # WB3a, WB3b
            continue
        if a == 'WSegSpace' and b == 'WSegSpace':
            continue                                        # WB3d
        if b in _SKIP:
            continue                                        # WB4
        # effective neighbours, skipping transparent characters
        p = i - 1
        while p > 0 and cls[p] in _SKIP and cls[p - 1] not in ('CR', 'LF', 'Newline'):
            p -= 1
        a = cls[p]
        pp = p - 1
        while pp > 0 and cls[pp] in _SKIP:
            pp -= 1
        before = cls[pp] if pp >= 0 and p > 0 else None
        q = i + 1
        while q < n and cls[q] in _SKIP:
            q += 1
        after = cls[q] if q < n else None
        if a in _AH and b in _AH:
            continue                                        # WB5
        if a in _AH and b in _MIDQ_LETTER and after in _AH:
            continue                                        # WB6
        if before in _AH and a in _MIDQ_LETTER and b in _AH:
            continue                                        # WB7
        if a == 'Hebrew_Letter' and b == 'Single_Quote':
            continue                                        # WB7a
        if a == 'Hebrew_Letter' and b == 'Double_Quote' and after == 'Hebrew_Letter':
            continue                                        # WB7b
        if before == 'Hebrew_Letter' and a == 'Double_Quote' and b == 'Hebrew_Letter':
            continue                                        # WB7c
        if a == 'Numeric' and b == 'Numeric':
            continue                                        # WB8
        if a in _AH and b == 'Numeric':
            continue                                        # WB9
        if a == 'Numeric' and b in _AH:
            continue                                        # WB10
        if before == 'Numeric' and a in _MIDQ_NUM and b == 'Numeric':
            continue                                        # WB11
        if a == 'Numeric' and b in _MIDQ_NUM and after == 'Numeric':
            continue                                        # WB12
        if a == 'Katakana' and b == 'Katakana':
            continue                                        # WB13
        if a in _AH + ('Numeric', 'Katakana', 'ExtendNumLet') and b == 'ExtendNumLet':
            continue                                        # WB13a
        if a == 'ExtendNumLet' and b in _AH + ('Numeric', 'Katakana'):
            continue                                        # WB13b
        if a == 'Regional_Indicator' and b == 'Regional_Indicator':
            k, run = p, 0
            while k >= 0 and (cls[k] == 'Regional_Indicator' or cls[k] in _SKIP):
                run += cls[k] == 'Regional_Indicator'
                k -= 1
            if run % 2 == 1:
                continue                                    # WB15, WB16: pairs of regional indicators
        breaks.append(i)                                    # WB999
    return breaks


def words(text):
    start = 0
    for i in word_boundaries(text):
        yield text[start:i]
        start = i
    if start < len(text):
        yield text[start:]


def _unwanted(c):
    """Whitespace, punctuation, marks, symbols, control and format characters (alignment.py:465-474)."""
    sub = unicodedata.category(c)
    return sub[0] in 'MPZS' or sub in ('Cc', 'Cf')


def splitwords(text):
    """Word tokens of a line: UAX #29 segments that contain at least one wanted character (alignment.py:476-486)."""
    return [w for w in words(text) if not all(_unwanted(c) for c in w)]
