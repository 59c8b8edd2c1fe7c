"""CPU oracle for the batched traffic-microsimulation hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package (``flow_amd``) may
import, call, link or execute anything under ``oracle/``.  The only permitted
users are ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg
of ``bench.py`` -- and there only as the checker / reported baseline, never as
the thing that is shipped or measured as the product.

What it is: a restatement, in numpy (``refsim.py`` closed loops, ``opennet.py``
open networks, ``controllers.py``, ``rewards.py``, ``network.py``) and plain C
(``csim/refsim.c``), of the
arithmetic that ``parthjaggi/flow`` performs on its ``Env.step`` path
(reference ``flow/envs/base.py:294-412``) -- the Flow-side Python controllers,
head-way bookkeeping, observation and reward functions -- plus an explicit
statement of the SUMO-side integration step that the reference delegates to a
third-party simulator which is NOT under ``/root/reference`` (Eclipse SUMO;
pins recorded in the reference: ``docs/source/flow_setup.rst:348`` commit
``2147d155b1``, ``scripts/setup_sumo_ubuntu1804.sh:15`` binaries
``flow-0.4.0``).

Parity status (see DESIGN.md "Oracle"):

* Flow-side arithmetic (controllers, fail-safes, rewards, placement,
  ``v_eq_max_function``): PINNED -- checked against the reference's own
  known-answer tests (``tests/fast_tests/test_controllers.py``,
  ``test_rewards.py``, ``test_environments.py``) and against golden vectors
  produced by importing the reference modules in the build container
  (``tests/golden/gen_golden.py``).
* SUMO-side integration (speed ramp of ``slowDown``, Euler update, wrap):
  pinned only to 2 decimals / 5 steps by the reference's emission fixture
  ``tests/fast_tests/test_files/ring_230_emission.csv``.  Finer than 1e-2 the
  SUMO boundary is PARITY UNPINNED (SUMO is not available in this image).
* Open networks (``opennet.py``): the Flow-side list logic of MergePOEnv /
  MultiAgentMergePOEnv / TraCIVehicle.update follows the source text and the
  reference's env tests (spaces, required parameters, observed ids) and its
  stored ``merge.json`` flow_params; the reference holds no numeric known answer
  for them.  The SUMO side (inflow insertion, arrivals, merge right of way:
  rules M1-M7) is PARITY UNPINNED.
"""
