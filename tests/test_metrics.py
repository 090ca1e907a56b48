"""Known answers for the evaluation metrics (cor_asv_ann_amd/metrics.py restating lib/alignment.py:140-486)."""
import math
import unicodedata

from cor_asv_ann_amd.metrics import Alignment, Edits, splitwords, words, normalize_text


def test_alignment_pairs_and_gaps():
    al = Alignment(0)
    al.set_seqs('abcd', 'abXd')
    assert al.get_best_alignment() == [('a', 'a'), ('b', 'b'), ('c', 'X'), ('d', 'd')]
    al.set_seqs('abcd', 'ad')                    # deletion: source symbols against the gap element
    assert al.get_best_alignment() == [('a', 'a'), ('b', 0), ('c', 0), ('d', 'd')]
    al.set_seqs('ad', 'abcd')
    assert al.get_best_alignment() == [('a', 'a'), (0, 'b'), (0, 'c'), ('d', 'd')]
    al.set_seqs('axyzd', 'apd')                  # replace block longer on the source side: pair, then gaps
    assert al.get_best_alignment() == [('a', 'a'), ('x', 'p'), ('y', 0), ('z', 0), ('d', 'd')]
    al.set_seqs('apd', 'axyzd')
    assert al.get_best_alignment() == [('a', 'a'), ('p', 'x'), (0, 'y'), (0, 'z'), ('d', 'd')]
    al.set_seqs(['the', 'quick', 'fox'], ['the', 'quik', 'brown', 'fox'])     # word lists: no recombination
    assert al.get_best_alignment() == [('the', 'the'), ('quick', 'quik'), (0, 'brown'), ('fox', 'fox')]


def test_combining_marks_join_their_base_letter():
    al = Alignment(0)
    # decomposed a-umlaut on the source side against a plain 'a': one symbol 'ä' vs 'a', not 'a'='a' + insertion
    al.set_seqs('bär', 'bar')
    assert al.get_best_alignment() == [('b', 'b'), ('ä', 'a'), ('r', 'r')]
    al.set_seqs('bar', 'bär')
    assert al.get_best_alignment() == [('b', 'b'), ('a', 'ä'), ('r', 'r')]
    al.set_seqs('bär', 'bär')
    assert al.get_best_alignment() == [('b', 'b'), ('ä', 'ä'), ('r', 'r')]
    # a mark after a non-letter stays a symbol of its own
    al.set_seqs('1̈x', '1x')
    assert al.get_best_alignment() == [('1', '1'), ('̈', 0), ('x', 'x')]
    # mark on the source side, different base symbol on the target side: the target symbol is left against a gap
    al.set_seqs('aéz', 'aXz')
    got = al.get_best_alignment()
    assert got in ([('a', 'a'), ('é', 'X'), ('z', 'z')], [('a', 'a'), ('é', 0), (0, 'X'), ('z', 'z')]), got


def test_adjusted_distance_and_normalisation():
    al = Alignment(0)
    assert al.get_adjusted_distance('abc\n', 'abd\n') == (1.0, 4)
    assert al.get_adjusted_distance('b\n', 'bb\n') == (1.0, 3)
    # historic_latin, GT level 1: long s = s, umlaut spellings, dash variants count as equal
    assert al.get_adjusted_distance('Waſſer', 'Wasser', normalization='historic_latin', gtlevel=1) == (0.0, 6)
    assert al.get_adjusted_distance('Waſſer', 'Wasser', normalization='historic_latin', gtlevel=2) == (2.0, 6)
    assert al.get_adjusted_distance('uͤber', 'über', normalization='historic_latin', gtlevel=1) == (0.0, 4)
    assert al.get_adjusted_distance('a—b', 'a-b', normalization='historic_latin') == (0.0, 3)
    assert al.get_adjusted_distance('a—b', 'a-b') == (1.0, 3)
    # level-2 replacements (ligatures, private-use code points) apply below GT level 3
    assert normalize_text('ﬁn  ā', 'historic_latin', 1) == 'fin ſſ ã'
    assert normalize_text('ﬁn', 'historic_latin', 3) == 'ﬁn'
    assert al.get_adjusted_distance('ﬁsch', 'fisch', normalization='historic_latin') == (0.0, 5)
    assert al.get_adjusted_distance('ﬁsch', 'fisch', normalization='historic_latin', gtlevel=3) == (2.0, 5)
    assert normalize_text('ä', 'NFC') == 'ä' and normalize_text(['ﬁ'], 'NFKC') == ['fi']
    # word level: token lists, equivalence applies symbol-wise only to strings that are equal as a whole
    assert al.get_adjusted_distance(['the', 'cat'], ['the', 'cot', 'sat']) == (2.0, 3)
    d, n, pairs = al.get_adjusted_distance('ab', 'b', return_alignment=True)
    assert (d, n) == (1.0, 2) and pairs == [('a', 0), ('b', 'b')]
    assert Alignment.get_levenshtein_distance('kitten', 'sitting') == (3, 7)
    assert Alignment.best_alignment('ab', 'ab') == [('a', 'a'), ('b', 'b')]


def test_confusion_table():
    al = Alignment(0, confusion=True)
    for src, tgt in (('abc\n', 'abd\n'), ('xbc\n', 'xbd\n'), ('m\n', 'rn\n'), ('tho\n', 'the\n')):
        al.get_adjusted_distance(src, tgt)
    table, total = al.get_confusion(2)
    assert table[0] == (2, ('c', 'd')) and len(table) == 2 and table[1][0] == 1
    # 'm' -> 'rn' aligns as m/r, gap/n: the gapped pair merges FORWARD into the next gap-free pair (here the newline),
    # giving a multi-character entry instead of a gap entry; among equal counts the entry seen later comes first
    al2 = Alignment(0, confusion=True)
    al2.get_adjusted_distance('am\n', 'arn\n')
    assert al2.get_confusion() == ([(1, ('\n', 'n\n')), (1, ('m', 'r'))], 2)
    # pairs that are equivalent under the normalisation are not counted
    al3 = Alignment(0, confusion=True)
    al3.get_adjusted_distance('ſo', 'so', normalization='historic_latin')
    assert al3.get_confusion() == ([], 0)
    import pytest
    with pytest.raises(Exception):
        Alignment(0).get_confusion()


def test_edits_running_statistics():
    e = Edits()
    e.add(1.0, 4, 'abc\n', 'abd\n')
    assert (e.length, e.mean, e.varia, e.steps) == (4, 0.25, 0.0, 1)
    e.add(0.0, 6, 'abcde\n', 'abcde\n')
    assert e.length == 10 and abs(e.mean - 0.1) < 1e-15 and abs(e.varia - 0.015) < 1e-15
    e.add(0.0, 0, '', '')                         # empty lines do not count
    assert e.length == 10 and e.steps == 2
    other = Edits()
    other.add(3.0, 5, 'x', 'y')
    e.merge(other, name_prefix='f/')
    assert e.length == 15 and abs(e.mean - 4.0 / 15) < 1e-12
    want_var = (4 * (0.25 - 4 / 15.) ** 2 + 6 * (4 / 15.) ** 2 + 5 * (0.6 - 4 / 15.) ** 2) / 15
    assert abs(e.varia - want_var) < 1e-12
    assert repr(e) == 'N=15 µ=%.2f σ²=%.2f' % (e.mean, e.varia)
    assert e.worst[0].length == 5 and e.worst[0].mean == 0.6       # worst line first
    h = Edits(histogram=True)
    h.add(1.0, 3, 'ab\n', 'ac\n')
    assert h.hist() == {'': (0, 0), '\n': (1, 1), 'a': (1, 1), 'b': (1, 0), 'c': (0, 1)}
    assert Edits().hist() == {}
    many = Edits()
    for k in range(30):
        many.add(float(k % 3), 10, 'x', 'y')
    assert len(many.worst) == 10                    # the worst 1 %, at least 10


def test_word_segmentation():
    assert splitwords("can't stop, 3.14 foo_bar\n") == ["can't", 'stop', '3.14', 'foo_bar']
    assert splitwords('»Daß« ſey\'s 1,000.5 a:b co-op') == ['Daß', "ſey's", '1,000.5', 'a:b', 'co', 'op']
    assert splitwords('e.g. U.S.A., etc.') == ['e.g', 'U.S.A', 'etc']
    assert splitwords('x2y 22nd 3a') == ['x2y', '22nd', '3a']
    assert splitwords('uͤber ähnlich') == ['uͤber', 'ähnlich']      # marks stay inside their word
    assert splitwords(' \t-- ... \n') == []
    assert list(words('ab  cd\r\nx')) == ['ab', '  ', 'cd', '\r\n', 'x']
    assert ''.join(words('Any text; re-joins: exactly!\n')) == 'Any text; re-joins: exactly!\n'
    assert splitwords('אב"ג カタカナ') == ['אב"ג', 'カタカナ']
    assert all(unicodedata.category(w[0])[0] in 'LN' for w in splitwords('Die 3 Haſen, und der Igel!'))


def test_historic_latin_normalisation_and_the_reference_quirk():
    """The reference pops the single-code-point replacements out of its global table during the first normalisation of a
    process (alignment.py:318-320), so only that first string gets them.  Here every call applies the whole table; the
    compatibility switch reproduces the reference's effective behaviour (cor_asv_ann_amd/metrics.py docstring)."""
    from cor_asv_ann_amd import metrics
    lig, plain = 'de\ufb01nire', 'definire'                    # U+FB01 LATIN SMALL LIGATURE FI: a single-code-point key
    assert metrics.normalize_text(lig, 'historic_latin') == plain
    assert metrics.normalize_text(lig, 'historic_latin') == plain                       # ... on every call
    a = metrics.Alignment(0)
    assert a.get_adjusted_distance(lig + '\n', plain + '\n', normalization='historic_latin')[0] == 0
    assert a.get_adjusted_distance(lig + '\n', plain + '\n', normalization='historic_latin')[0] == 0
    metrics.reference_quirks(True)
    try:
        assert metrics.normalize_text(lig, 'historic_latin') == plain                   # the first string of the "process"
        assert metrics.normalize_text(lig, 'historic_latin') == lig                     # popped: later strings keep the ligature
        assert a.get_adjusted_distance(lig + '\n', plain + '\n', normalization='historic_latin')[0] > 0
    finally:
        metrics.reference_quirks(False)
    assert metrics.normalize_text(lig, 'historic_latin') == plain
