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


# ---- the remaining scheduler_test.go scenarios -------------------------------------------------
def test_multiple_preferences_scale_up():
    sc.scenario_multiple_preferences_scale_up(factory)


def test_faulty_node_spec_version():
    sc.scenario_faulty_node_spec_version(factory)


@pytest.mark.parametrize("with_generic", [False, True])
def test_resource_constraint_ha(with_generic):
    sc.scenario_resource_constraint_ha(factory, with_generic)


@pytest.mark.parametrize("with_generic", [False, True])
def test_resource_constraint_dead_task(with_generic):
    sc.scenario_resource_constraint_dead_task(factory, with_generic)


@pytest.mark.parametrize("with_generic", [False, True])
def test_preexisting_dead_task(with_generic):
    sc.scenario_preexisting_dead_task(factory, with_generic)


def test_unassigned_map():
    sc.scenario_unassigned_map(factory)


def test_preassigned_tasks():
    sc.scenario_preassigned_tasks(factory)


def test_ignore_tasks():
    sc.scenario_ignore_tasks(factory)


def test_unscheduleable_task():
    sc.scenario_unscheduleable_task(factory)


def test_plugin_constraint():
    sc.scenario_plugin_constraint(factory)
