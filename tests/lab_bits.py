"""Variant words of the tests: ONE integer that names a product tuning (``DDViewBatch.tuning``, ``DD_TUNE_*`` of include/ddcore.h)
together with experiment switches (``DD_LAB_*`` of include/ddcore_lab.h, a thread-local debug word since ABI 14), so that a test
can sweep "every way the library can run this batch" with one parameter.  The positions of the switch bits here are the tests' own
(those the bits had inside ``tuning`` up to ABI 13); ``split`` turns a word into what the binding takes."""

from depthdensifier_amd import _lib

FAULT = 64                 # fault injection: an in-kernel scan gives up
LIST_ORDER = 32            # rows in list order (no shift of the wave runs onto 128-byte lines)
W32, W64 = 2 << 20, 3 << 20        # polling lanes of the decoupled look-back
CLASSIC = 1 << 26          # the decoupled look-back of rounds 1-4 instead of the scan service

_SWITCHES = ((FAULT, _lib.DD_LAB_FAULT_INJECT), (LIST_ORDER, _lib.DD_LAB_LIST_ORDER), (CLASSIC, _lib.DD_LAB_LOOKBACK))


def split(word: int) -> tuple:
    """variant word -> (tuning, lab)"""
    word = int(word)
    lab = 0
    for bit, sw in _SWITCHES:
        if word & bit:
            lab |= sw
            word &= ~bit
    lanes = (word >> 20) & 3
    lab |= _lib.DD_LAB_POLL_LANES_64 if lanes == 3 else _lib.DD_LAB_POLL_LANES_32 if lanes == 2 else 0
    word &= ~(3 << 20)
    return word, lab


def kw(word: int) -> dict:
    t, l = split(word)
    return {"tuning": t, "lab": l}


def set_on(batch, word: int):
    batch.tuning, batch.lab = split(word)
    return batch
