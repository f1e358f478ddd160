"""The callers' post-processing: token collapse for H graphs, edit
distance and word error rate."""
from kaldi_decoder import scoring


def test_edit_distance_known_answers():
    assert scoring.edit_distance([], []) == 0
    assert scoring.edit_distance([1, 2, 3], []) == 3
    assert scoring.edit_distance([], [4, 5]) == 2
    assert scoring.edit_distance("kitten", "sitting") == 3
    assert scoring.edit_distance("intention", "execution") == 5
    assert scoring.edit_distance([1, 2, 3, 4], [1, 3, 4]) == 1       # one deletion
    assert scoring.edit_distance([1, 2, 3], [1, 9, 2, 3]) == 1       # one insertion
    assert scoring.edit_distance([1, 2, 3], [1, 7, 3]) == 1          # one substitution
    a, b = [5, 1, 4, 1, 5, 9, 2, 6], [3, 1, 4, 1, 5, 9, 2, 6, 5]
    assert scoring.edit_distance(a, b) == scoring.edit_distance(b, a) == 2


def test_ctc_collapse():
    # ilabel = token + 1, 0 = epsilon; token 0 = blank
    assert scoring.ctc_collapse([]) == []
    assert scoring.ctc_collapse([1, 1, 1]) == []                       # blanks only
    assert scoring.ctc_collapse([3, 3, 1, 3]) == [2, 2]                # a a _ a -> a a
    assert scoring.ctc_collapse([3, 3, 3, 4, 4, 1, 1, 5]) == [2, 3, 4]
    assert scoring.ctc_collapse([0, 3, 0, 3, 1]) == [2]                # epsilons vanish first
    assert scoring.ctc_collapse([2, 1, 2, 2, 1, 1, 2]) == [1, 1, 1]


def test_word_error_rate():
    wer, errs, words = scoring.word_error_rate([[1, 2, 3], [4, 5]], [[1, 2, 3], [4, 6, 5]])
    assert (errs, words) == (1, 5) and abs(wer - 0.2) < 1e-12
    assert scoring.word_error_rate([], []) == (0.0, 0, 0)
