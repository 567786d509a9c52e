"""bench.py prints its ONE JSON line even when an extra ensemble shape stalls: the shapes run last, under a watchdog that prints the
headline part with whatever shapes finished and ends the process (bench.finish_line).  No GPU here: the shapes are stand-ins."""
import json
import threading
import time

import bench


def _line():
    return dict(metric='REMD iterations/s', value=12.0, unit='iterations/s', shapes=None)


def test_line_is_printed_once_after_the_shapes():
    got = []
    def shapes(d):
        d['strong128_alanine'] = dict(value=14.0)
    bench.finish_line(_line(), 0, shapes, 5.0, write=got.append, end_process=lambda: got.append('END'))
    assert len(got) == 1
    line = json.loads(got[0])
    assert line['value'] == 12.0 and line['shapes'] == {'strong128_alanine': {'value': 14.0}}


def test_no_shapes_and_other_ranks():
    got = []
    bench.finish_line(_line(), 0, None, 5.0, write=got.append)
    assert len(got) == 1 and json.loads(got[0])['shapes'] is None
    got = []
    bench.finish_line(None, 1, lambda d: None, 5.0, write=got.append)            # a rank other than 0 prints nothing
    assert got == []


def test_a_stalled_shape_does_not_take_the_line_down():
    got = []
    ended = threading.Event()
    release = threading.Event()
    def shapes(d):
        d['strong128_alanine'] = dict(value=14.0)                                  # this one finished
        release.wait(10.0)                                                         # the next one hangs in a device call
    def end():
        got.append('END'); ended.set(); release.set()                              # (the real one is os._exit(0))
    t0 = time.perf_counter()
    bench.finish_line(_line(), 0, shapes, 0.3, write=got.append, end_process=end)
    assert ended.is_set() and time.perf_counter() - t0 < 5.0
    assert got[-1] == 'END' and len(got) == 2                                      # one line, then the end of the process; nothing after
    line = json.loads(got[0])
    assert line['value'] == 12.0 and line['shapes']['strong128_alanine'] == {'value': 14.0}
    assert 'did not finish' in line['shapes']['watchdog']


def test_an_exception_in_the_shapes_is_reported_in_the_line():
    got = []
    def shapes(d):
        d['strong128_alanine'] = dict(value=14.0)
        raise RuntimeError('device gone')
    bench.finish_line(_line(), 0, shapes, 5.0, write=got.append, end_process=lambda: got.append('END'))
    assert len(got) == 1
    line = json.loads(got[0])
    assert line['value'] == 12.0 and line['shapes']['strong128_alanine'] == {'value': 14.0} and 'device gone' in line['shapes']['error']
