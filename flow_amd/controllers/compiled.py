"""A user-defined acceleration controller.

The reference's extension point is a Python subclass of ``BaseController`` whose ``get_accel(env)`` runs once per vehicle
and step (flow/controllers/base_controller.py:42-118).  Here the step loop is a HIP kernel, so a user controller's
``get_accel`` is a device function: ``CompiledController`` carries its body as C++ text, ``flow_amd.build.build_user``
compiles it into a copy of libflowsim.so (cached under flow_amd/_user/) and every handle whose vehicles use it steps on
that copy (controller id FS_CTRL_USER, generic step kernels).  Acceleration noise, fail-safes, the speed modes and the
junction rule apply to it exactly as to the built-in controllers.

    class TimeGap(CompiledController):
        SOURCE = '''
            const T gap_err = h - p[0] * v - p[1];
            const T a = p[2] * gap_err + p[3] * (v_lead - v);
            return has_lead ? tmin(tmax(a, -max_accel), max_accel) : max_accel;
        '''
        def __init__(self, veh_id, car_following_params, t_gap=1.2, s0=2.0, k1=0.3, k2=0.6, **kw):
            CompiledController.__init__(self, veh_id, car_following_params, params=[t_gap, s0, k1, k2], **kw)

    vehicles.add("human", acceleration_controller=(TimeGap, {"t_gap": 1.0}), num_vehicles=22)

In scope of the body: ``v``, ``v_lead``, ``h`` (bumper-to-bumper headway; 1000 without a leader), ``has_lead``,
``v_follow``, ``h_follow``, ``dt``, ``max_accel``, ``p[0..7]``; ``T`` is ``float`` or ``double``; helpers ``tmin``, ``tmax``,
``tabs``, ``tsqrt``.  All user controllers of one environment must share ONE body (one library); their parameters may
differ per vehicle type."""
from flow_amd import _lib as L
from flow_amd.controllers.base_controller import BaseController


class CompiledController(BaseController):
    FS_ID = L.FS_CTRL_USER
    SOURCE = None                      # the body of get_accel (C++), given by the subclass or per instance

    def __init__(self, veh_id, car_following_params, params=(), source=None, delay=0, fail_safe=None, noise=0):
        BaseController.__init__(self, veh_id, car_following_params, delay=delay, fail_safe=fail_safe, noise=noise)
        self.source = source if source is not None else type(self).SOURCE
        if not self.source or not str(self.source).strip():
            raise ValueError("CompiledController needs the C++ body of its get_accel (SOURCE / source=)")
        self.params = [float(x) for x in params]
        if len(self.params) > 8:
            raise ValueError("a compiled controller takes at most 8 parameters")

    def fs_params(self):
        return list(self.params)
