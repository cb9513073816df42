"""Mirror of halo2's `SimpleFloorPlanner` (halo2_proofs v2023_04_20 src/circuit/floor_planner/single_pass.rs):
the layouter zero_g's `WnnCircuit` declares (/root/reference/src/gadgets/wnn.rs:321-327).

    region start  = max over the columns the region touches (advice columns AND selectors) of that
                    column's next free row; afterwards each touched column's next free row becomes
                    start + row_count of the region;
    constants     = `assign_advice_from_constant` / `constrain_constant` queue (value, cell); after the
                    region they go, one row each, into the first `enable_constant` fixed column with a
                    copy constraint to the cell;
    tables        = `assign_table` fills fixed columns from row 0 and pads the rest of the usable rows
                    with the first row's value (SimpleTableLayouter default value);
    instances     = `constrain_instance` is a copy constraint to the instance column.

Values are canonical integers mod r.  Works on circuit.ConstraintSystem / circuit.Assignment.
"""
from __future__ import annotations

from circuit import ADVICE, FIXED, INSTANCE, R, Assignment, ConstraintSystem


class AssignedCell:
    __slots__ = ("region", "column", "offset", "value")

    def __init__(self, region, column, offset, value):
        self.region, self.column, self.offset, self.value = region, column, offset, value % R

    @property
    def row(self) -> int:
        return self.region.start + self.offset

    def cell(self):
        return (ADVICE, self.column, self.row)

    def copy_advice(self, region: "Region", column: int, offset: int) -> "AssignedCell":
        """AssignedCell::copy_advice: assign the same value, constrain equal."""
        c = region.assign_advice(column, offset, self.value)
        region.copies.append((self, c))
        return c


class Region:
    def __init__(self, layouter: "Layouter"):
        self.layouter = layouter
        self.start = None
        self.cells = []       # AssignedCell (advice)
        self.selectors = []   # (selector id, offset)
        self.constants = []   # (value, AssignedCell)
        self.copies = []      # (source AssignedCell of an earlier region or this one, AssignedCell)

    def assign_advice(self, column: int, offset: int, value: int) -> AssignedCell:
        c = AssignedCell(self, column, offset, value)
        self.cells.append(c)
        return c

    def assign_advice_from_constant(self, column: int, offset: int, constant: int) -> AssignedCell:
        c = self.assign_advice(column, offset, constant)
        self.constants.append((constant % R, c))
        return c

    def constrain_constant(self, cell: AssignedCell, constant: int):
        self.constants.append((constant % R, cell))

    def enable_selector(self, selector: int, offset: int):
        self.selectors.append((selector, offset))


class Layouter:
    def __init__(self, cs: ConstraintSystem, constants_column: int):
        self.cs = cs
        self.asg = Assignment(cs)
        self.constants_column = constants_column
        self.columns = {}      # ("a", col) | ("s", selector) | ("f", col) -> next free row
        self.n_regions = 0
        self.table_columns = set()

    def assign_region(self, fn):
        """fn(region) -> result; cells are placed once the region's shape is known."""
        region = Region(self)
        result = fn(region)
        used = {("a", c.column) for c in region.cells} | {("s", s) for s, _ in region.selectors}
        rows = 1 + max([c.offset for c in region.cells] + [o for _, o in region.selectors], default=-1)
        region.start = max((self.columns.get(u, 0) for u in used), default=0)
        for u in used:
            self.columns[u] = region.start + rows
        assert region.start + rows <= self.cs.usable_rows(), "not enough rows available (k too small)"
        self.commit_region(region)
        self.n_regions += 1
        return result

    def commit_region(self, region: Region):
        """the placed region into the assignment (witness_tape.TapeLayouter records cells instead)"""
        asg = self.asg
        for c in region.cells:
            asg.set(ADVICE, c.column, c.row, c.value)
        for s, off in region.selectors:
            asg.selectors[s].add(region.start + off)
        for src, dst in region.copies:
            asg.copy(src.cell(), dst.cell())
        key = ("f", self.constants_column)
        for value, cell in region.constants:
            row = self.columns.get(key, 0)
            assert row < self.cs.usable_rows(), "constants column is full"
            asg.set(FIXED, self.constants_column, row, value)
            asg.copy((FIXED, self.constants_column, row), cell.cell())
            self.columns[key] = row + 1

    def assign_table(self, columns, rows):
        """columns: fixed column ids; rows: list of tuples, one value per column."""
        usable = self.cs.usable_rows()
        assert 0 < len(rows) <= usable, "table does not fit"
        for j, col in enumerate(columns):
            assert col not in self.table_columns, "table column assigned twice"
            self.table_columns.add(col)
            dst = self.asg.fixed[col]
            for i, r in enumerate(rows):
                dst[i] = r[j] % R
            for i in range(len(rows), usable):
                dst[i] = rows[0][j] % R

    def constrain_instance(self, cell: AssignedCell, instance_column: int, row: int, value: int):
        self.asg.set(INSTANCE, instance_column, row, value)
        self.asg.copy(cell.cell(), (INSTANCE, instance_column, row))

    def rows_used(self) -> int:
        return max(self.columns.values(), default=0)
