"""placeholder -- replaced below"""


class BatchedEditEval:
    @staticmethod
    def supports(editor, eval_data, edit_n):
        return False
