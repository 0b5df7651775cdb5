# -*- coding: utf-8 -*-
'''
Implicit Euler for problems given in the  alpha*M*u + beta*F(u, t)  protocol.
The reference drives `flow.heat.Heat` with the third-party
`parabolic.ImplicitEuler(problem).step(u0, t, dt)`
(tests/test_boussinesq.py:220-229), which is not available offline; this is
the same one-liner: solve  M u1 - dt F(t+dt, u1) = M u0.
'''


class ImplicitEuler(object):
    order = 1.0

    def __init__(self, problem):
        self.problem = problem

    def step(self, u0, t, dt):
        rhs = self.problem.eval_alpha_M_beta_F(1.0, 0.0, u0, t)   # M u0 (+0)
        # eval adds beta*b = 0; the solve applies the Dirichlet conditions
        return self.problem.solve_alpha_M_beta_F(1.0, -dt, rhs, t + dt)
