"""Pins the CPU oracle against the reference's own scheduler tests (scheduler_test.go), CPU only."""
import pytest

import orc
import scenarios as sc


def factory():
    return orc.Oracle()


def test_basic():
    sc.scenario_basic(factory)


@pytest.mark.parametrize("use_spec_version", [False, True])
def test_ha(use_spec_version):
    sc.scenario_ha(factory, use_spec_version)


@pytest.mark.parametrize("use_spec_version", [False, True])
def test_preferences(use_spec_version):
    sc.scenario_preferences(factory, use_spec_version)


def test_no_ready_nodes():
    sc.scenario_no_ready_nodes(factory)


@pytest.mark.parametrize("with_generic", [False, True])
def test_resource_constraint(with_generic):
    sc.scenario_resource_constraint(factory, with_generic)


def test_platform():
    sc.scenario_platform(factory)


def test_host_port():
    sc.scenario_host_port(factory)


def test_max_replicas():
    sc.scenario_max_replicas(factory)


def test_faulty_node():
    sc.scenario_faulty_node(factory)


@pytest.mark.parametrize("use_spec_version", [False, True])
@pytest.mark.parametrize("with_generic", [False, True])
def test_multiple_preferences(use_spec_version, with_generic):
    sc.scenario_multiple_preferences(factory, use_spec_version, with_generic)
