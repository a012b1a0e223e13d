"""The reference's own known-answer tests for the path: the 24 #[test]s of
raytracer/src/vec3.rs:425-564, replayed against the oracle's Vec3 (values are the
reference's test data; this is what PINS the oracle's Vec3 arithmetic)."""
import math

import numpy as np
import pytest

import oracle as O

ADD, ADDS, SUB, SUBS, DOT, MULS, DIV, ELEMUL, CROSS, NEG, SQLEN, LEN, UNIT = range(13)


def eq(a, b):
    assert np.array_equal(np.asarray(a, dtype=float), np.asarray(b, dtype=float)), (a, b)


def test_new():  # vec3.rs:429-431
    eq(O.vec3_op(ADDS, (1.0, 2.0, 3.0), s=0.0), (1.0, 2.0, 3.0))


def test_add_and_add_assign():  # :433-447
    eq(O.vec3_op(ADD, (1.0, 0.0, -1.0), (2.0, 4.0, 6.0)), (3.0, 4.0, 5.0))


def test_add_f64_and_add_assign_f64():  # :449-463
    eq(O.vec3_op(ADDS, (1.0, 0.0, -1.0), s=233.0), (234.0, 233.0, 232.0))


def test_sub_and_sub_assign():  # :465-479
    eq(O.vec3_op(SUB, (1.0, 0.0, -1.0), (2.0, 4.0, 6.0)), (-1.0, -4.0, -7.0))


def test_sub_f64_and_sub_assign_f64():  # :481-491
    eq(O.vec3_op(SUBS, (1.0, 0.0, -1.0), s=1.0), (0.0, -1.0, -2.0))


def test_mul_is_dot():  # :493-496 (quirk Q1)
    assert O.vec3_op(DOT, (1.0, 0.0, -1.0), (1.0, 1.0, 1.0))[0] == 0.0


def test_mul_assign_and_mul_f64():  # :498-508
    eq(O.vec3_op(MULS, (1.0, 0.0, -1.0), s=2.0), (2.0, 0.0, -2.0))
    eq(O.vec3_op(MULS, (1.0, 0.0, -1.0), s=1.0), (1.0, 0.0, -1.0))


def test_div():  # :510-516
    eq(O.vec3_op(DIV, (1.0, -2.0, 0.0), s=2.0), (0.5, -1.0, 0.0))


def test_elemul():  # :518-524
    eq(O.vec3_op(ELEMUL, (1.0, 2.0, 3.0), (1.0, 2.0, 3.0)), (1.0, 4.0, 9.0))


def test_cross():  # :526-532
    eq(O.vec3_op(CROSS, (1.0, 2.0, 3.0), (2.0, 3.0, 4.0)), (8.0 - 9.0, 6.0 - 4.0, 3.0 - 4.0))


def test_neg():  # :534-537
    eq(O.vec3_op(NEG, (1.0, -2.0, 3.0)), (-1.0, 2.0, -3.0))


def test_squared_length():  # :539-542
    assert O.vec3_op(SQLEN, (1.0, 2.0, 3.0))[0] == 14.0


def test_length():  # :544-550
    assert O.vec3_op(LEN, (3.0, 4.0, 5.0))[0] == math.sqrt(3.0 * 3.0 + 4.0 * 4.0 + 5.0 * 5.0)


def test_unit():  # :552-559
    eq(O.vec3_op(UNIT, (233.0, 0.0, 0.0)), (1.0, 0.0, 0.0))
    eq(O.vec3_op(UNIT, (-233.0, 0.0, 0.0)), (-1.0, 0.0, 0.0))


def test_unit_panic():  # :561-564 #[should_panic] -> error code
    with pytest.raises(O.OracleError):
        O.vec3_op(UNIT, (0.0, 0.0, 0.0))
