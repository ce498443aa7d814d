"""CPU: the host layer's JSON reader and writer (swarmkit_amd/csrc/swp_json.hpp) against Python's json module on random documents —
nesting, every kind of string (escapes, control characters, two- to four-byte UTF-8, surrogate pairs written as escapes), integers over
the whole int64 range and above it, reals, empty containers, white space: what the C++ side reads and writes back must load as the same
document; and text Python refuses must be refused."""
import json
import os
import random
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "cxx", "json_echo.cpp")
DEPS = [SRC, os.path.join(HERE, "..", "swarmkit_amd", "csrc", "swp_json.hpp")]


@pytest.fixture(scope="module")
def echo():
    out = os.path.join(HERE, "_build", "json_echo")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in DEPS):
        tmp = out + ".%d.tmp" % os.getpid()
        subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-Wall", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-o", tmp, SRC], check=True)
        os.replace(tmp, out)

    def run(lines):
        r = subprocess.run([out], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1"))
        assert r.returncode == 0, r.stderr[-3000:]
        got = r.stdout.split("\n")[:-1]
        assert len(got) == len(lines)
        return got
    return run


ALPHABET = list("abcXYZ019 _-./:=") + ['"', "\\", "\n", "\t", "\r", "\b", "\f", "\x01", "\x1f", "\x7f", "é", "ß", "中", "€", "\U0001F600", "\U00010000", "￿", "/"]


def rand_string(rng):
    return "".join(rng.choice(ALPHABET) for _ in range(rng.choice([0, 1, 3, 9, 15, 16, 17, 40])))


def rand_doc(rng, depth=0):
    k = rng.random()
    if depth > 5 or k < 0.35:
        c = rng.randrange(9)
        if c == 0:
            return None
        if c == 1:
            return rng.random() < 0.5
        if c == 2:
            return rng.choice([0, 1, -1, 7, 250000000, 2**31, 2**53 + 1, 2**63 - 1, -2**63, 10**17 - 1, 10**17, 10**18, -10**18, rng.randrange(-2**63, 2**63)])
        if c == 3:
            return rng.choice([0.5, -1.25, 1e-9, 3.141592653589793, 1e100, -2.5e-300, float(rng.randrange(1000)) / 7.0])
        return rand_string(rng)
    if k < 0.7:
        return {rand_string(rng)[:12] or "k": rand_doc(rng, depth + 1) for _ in range(rng.randrange(0, 6))}
    return [rand_doc(rng, depth + 1) for _ in range(rng.randrange(0, 6))]


def as_the_reader_sees_it(d):
    """an integer member above int64 keeps its bit pattern (MaxReplicas is a uint64): the writer prints it as the int64 it is stored as"""
    return d


@pytest.mark.parametrize("seed", range(6))
def test_reader_and_writer_agree_with_python(echo, seed):
    rng = random.Random(0xD0C + seed)
    docs = [rand_doc(rng) for _ in range(400)]
    texts = []
    for d in docs:
        style = rng.randrange(4)
        if style == 0:
            texts.append(json.dumps(d))                                        # \\u escapes for everything outside ASCII (surrogate pairs included)
        elif style == 1:
            texts.append(json.dumps(d, ensure_ascii=False))                    # raw UTF-8
        elif style == 2:
            texts.append(json.dumps(d, separators=(",", ":")))
        else:
            texts.append(json.dumps(d, indent=rng.choice([1, 3])).replace("\n", rng.choice([" ", "\t", "\r"])))   # white space everywhere
    got = echo(texts)
    for d, t, g in zip(docs, texts, got):
        assert not g.startswith("!"), (t, g)
        assert json.loads(g) == d, (t, g)


def test_uint64_above_int64_keeps_its_bits(echo):
    got = echo(['{"MaxReplicas":18446744073709551615}', "[9223372036854775808]", "[123456789012345678901234567890]"])
    assert json.loads(got[0]) == {"MaxReplicas": -1} and json.loads(got[1]) == [-2**63]   # the int64 the uint64 is stored as
    assert json.loads(got[2]) == [1.2345678901234568e+29]                                 # beyond uint64: a real


BAD = ['{"a":1', '{"a":1}}', '[1,2,]', '{"a" 1}', '{a:1}', '"abc', '"a\\x"', '"\\u12"', "nul", "tru", "", "[1 2]", '{"a":1,}', "-", "+1", ".5", "[01x]", '["a"', "{", "[",
       '{"a":{"b":[{"c":', "1 2", "[1]]", '{"a":}', "[,1]", '"a" "b"', "\\", "'a'", "[1,,2]", "{\"a\":1 \"b\":2}"]


def test_text_that_is_not_json_is_refused(echo):
    for t in BAD:
        with pytest.raises(ValueError):
            json.loads(t)
    for t, g in zip(BAD, echo(BAD)):
        assert g.startswith("!json: "), (t, g)


def test_the_two_deliberate_leniencies(echo):
    """Control characters inside a string are read (Python's strict mode refuses them; Go's decoder refuses them too — the writer escapes
    them again, so nothing that is not JSON ever comes out); half a surrogate pair written as an escape reads as U+FFFD, as in Go's
    encoding/json (Python keeps the lone surrogate)."""
    got = echo(['"a' + chr(1) + 'b"', '"' + chr(92) + 'ud800' + chr(92) + 'u0041x"', '"' + chr(92) + 'udc00"'])
    assert json.loads(got[0]) == "a\x01b"
    assert json.loads(got[1]) == chr(0xFFFD) + "Ax"
    assert json.loads(got[2]) == chr(0xFFFD)
