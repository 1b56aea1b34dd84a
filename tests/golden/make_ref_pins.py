#!/usr/bin/env python3
"""Build-container script: pins the repo's re-typed inputs to what the reference HOLDS.

The reference has no tests or golden vectors for the solve path and cannot be imported here
(forcespro / casadi / forwardkinematics / urdfenvs absent), but it does hold data:
three YAML configs, three URDFs and the scenario constants typed into its example programs.
This script reads those files AS DATA (yaml / xml / python-literal parsing, nothing is
imported or executed) and writes ``tests/golden/ref_pins.json``:

    configs   examples/config/{pointRobot,boxer,panda}Mpc.yaml          -> mpc / robot blocks
              ros_bridge/src/robotmpcs_ros/config/boxer_mpc_config.yaml (no ``objectives`` key)
    chains    examples/assets/*/*.urdf -> every joint: type, parent, child, origin xyz / rpy, axis
    scenes    examples/pointRobot_example.py:31-65, panda_example.py:53-83,
              boxer_example.py:46-93 -> r_body, limits, limits_u, goal, obstacles, start velocity

Run in the build container only (``/root/reference`` does not exist on the GPU box); the JSON
is the committed fixture, ``tests/test_reference_pins.py`` checks the repo against it.
"""
import ast
import json
import os
import sys
import xml.etree.ElementTree as ET

import yaml

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_pins.json")


def urdf_joints(path):
    robot = ET.parse(path).getroot()
    out = []
    for j in robot.findall("joint"):
        o, a = j.find("origin"), j.find("axis")
        out.append({
            "name": j.get("name"), "type": j.get("type"),
            "parent": j.find("parent").get("link"), "child": j.find("child").get("link"),
            "xyz": [float(v) for v in (o.get("xyz") if o is not None and o.get("xyz") else "0 0 0").split()],
            "rpy": [float(v) for v in (o.get("rpy") if o is not None and o.get("rpy") else "0 0 0").split()],
            "axis": [float(v) for v in a.get("xyz").split()] if a is not None else None,
        })
    return {"links": [l.get("name") for l in robot.findall("link")], "joints": out}


class _Lit(ast.NodeVisitor):
    """Collects ``self._x = <literal>`` / ``name = <literal>`` assignments; ``np.array(L)`` counts as L,
    ``np.pi`` as pi."""

    def __init__(self):
        self.vals = {}
        self._depth = 0

    def _ev(self, node):
        import math
        if isinstance(node, ast.Call) and getattr(node.func, "attr", "") == "array" and node.args:
            return self._ev(node.args[0])
        if isinstance(node, ast.Attribute) and node.attr == "pi":
            return math.pi
        if isinstance(node, (ast.List, ast.Tuple, ast.Dict)):
            self._depth += 1
            try:
                if isinstance(node, ast.Dict):
                    return {self._ev(k): self._ev(v) for k, v in zip(node.keys, node.values)}
                return [self._ev(e) for e in node.elts]
            finally:
                self._depth -= 1
        if isinstance(node, ast.UnaryOp) and isinstance(node.op, ast.USub):
            return -self._ev(node.operand)
        if isinstance(node, ast.BinOp) and isinstance(node.op, (ast.Div, ast.Mult, ast.Add, ast.Sub)):
            a, b = self._ev(node.left), self._ev(node.right)
            return {ast.Div: a / b, ast.Mult: a * b, ast.Add: a + b, ast.Sub: a - b}[type(node.op)]
        if isinstance(node, ast.Constant):
            return node.value
        if self._depth > 0:
            return "<expr>"   # non-literal entry inside a container (e.g. "child_link": self._n)
        raise ValueError("not a literal")

    def visit_Assign(self, node):
        for t in node.targets:
            name = t.attr if isinstance(t, ast.Attribute) else (t.id if isinstance(t, ast.Name) else None)
            if name is None:
                continue
            try:
                self.vals.setdefault(name, []).append(self._ev(node.value))
            except (ValueError, TypeError, KeyError, AttributeError):
                pass
        self.generic_visit(node)


def example_literals(path):
    v = _Lit()
    v.visit(ast.parse(open(path).read()))
    return v.vals


def main():
    pins = {"source": "maxspahn/robot_mpcs v0.2.1 (files read as data by tests/golden/make_ref_pins.py)",
            "configs": {}, "chains": {}, "scenes": {}}
    for robot in ("pointRobot", "boxer", "panda"):
        cfg = yaml.safe_load(open(os.path.join(REF, "examples", "config", robot + "Mpc.yaml")))
        pins["configs"][robot] = cfg
        pins["chains"][robot] = urdf_joints(os.path.join(REF, "examples", "assets", robot, cfg["robot"]["urdf_file"]))
    pins["configs"]["boxer_ros"] = yaml.safe_load(
        open(os.path.join(REF, "ros_bridge", "src", "robotmpcs_ros", "config", "boxer_mpc_config.yaml")))

    lit = example_literals(os.path.join(REF, "examples", "pointRobot_example.py"))
    pins["scenes"]["pointRobot"] = {
        "r_body": lit["_r_body"][0], "limits": lit["_limits"][0], "limits_u": lit["_limits_u"][0],
        "obstacles": [lit["static_obst_dict"][0]["geometry"]],
        "goal": lit["goal_dict"][0]["subgoal0"]["desired_position"],
        "vel0": lit["vel0"][0],
    }
    lit = example_literals(os.path.join(REF, "examples", "panda_example.py"))
    pins["scenes"]["panda"] = {
        "r_body": lit["_r_body"][0], "limits": lit["_limits"][0], "limits_u": lit["_limits_u"][0],
        "obstacles": [lit["static_obst_dict"][0]["geometry"]],
        "goal": lit["goal_dict"][0]["subgoal0"]["desired_position"],
    }
    lit = example_literals(os.path.join(REF, "examples", "boxer_example.py"))
    pins["scenes"]["boxer"] = {
        "r_body": lit["_r_body"][0], "limits": lit["_limits"][0], "limits_u": lit["_limits_u"][0],
        "obstacles": [lit["obstacle_1_dict"][0]["geometry"], lit["obstacle_2_dict"][0]["geometry"]],
        "goal": lit["goal_dict"][0]["subgoal0"]["desired_position"],
    }
    with open(OUT, "w") as f:
        json.dump(pins, f, indent=1, sort_keys=True)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
