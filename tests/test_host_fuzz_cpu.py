"""CPU: the host mirror's string handling (constraint.Parse, key folding) against the oracle's restatement of the Go
behaviour on seeded random expressions — the part of the drop-in that a cgo shim gets for free from the Go standard
library and that the Python test bed has to reproduce."""
import random

import pytest

import pyhost

import orc
from swarmkit_amd import host as swhost

ALPHA = list("abcXYZ019_-.") + ["node", "labels", "engine", "==", "!=", " ", "\t", "=", "!", "K", "K", "ſ", "/", ":", "*", "(", ")", "é", "", "id", "ip"]


def rand_expr(rng):
    kind = rng.random()
    if kind < 0.5:
        key = rng.choice(["node.id", "node.hostname", "Node.Labels.zone", "engine.labels.x", "node.role", "node.platform.os", "NODE.IP", "node.labels.", "x"])
        op = rng.choice(["==", "!=", " == ", "!= ", "=", "==="])
        val = "".join(rng.choice(ALPHA) for _ in range(rng.randrange(0, 5)))
        return key + op + val
    return "".join(rng.choice(ALPHA) for _ in range(rng.randrange(0, 9)))


@pytest.mark.parametrize("seed", range(8))
def test_parse_constraints_matches_the_oracle(seed):
    rng = random.Random(0xFADE + seed)
    for _ in range(400):
        exprs = [rand_expr(rng) for _ in range(rng.randrange(1, 4))]
        want, err = orc.constraint_parse(exprs)
        got = pyhost.parse_constraints(exprs)
        if want is None:
            assert got is None, (exprs, err, got)
        else:
            assert got is not None, (exprs, want)
            assert [(k, o, v) for k, o, v in got] == [(k, o, v) for k, o, v in want], exprs


def test_fold_eq_matches_equal_fold_on_key_alphabet():
    rng = random.Random(5)
    chars = list("abkKsS.-_09") + ["K", "ſ"]
    for _ in range(3000):
        a = "".join(rng.choice(chars) for _ in range(rng.randrange(0, 6)))
        b = "".join(rng.choice(chars) for _ in range(rng.randrange(0, 6)))
        if rng.random() < 0.5:
            b = "".join(rng.choice([c, c.upper(), c.lower()]) for c in a)
        assert pyhost._fold_eq(a, b) == orc.equal_fold(a, b), (a, b)
